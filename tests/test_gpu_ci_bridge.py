"""GPU: the standard <-> conjugate-invariant bridges of ring/conjugate_invariant.go (callers: schemes/ckks/bridge.go:82-83, 116-117,
core/rlwe/keygenerator.go:213) against literal restatements of the reference's loops (oracle/compose.py); the reference holds no vector
for these (parity pinned by the restatement of the loops alone)."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod

pytestmark = pytest.mark.gpu


def _block(rng, mods, B, N):
    return np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(B)])


@pytest.mark.parametrize("n", [8, 64, 4096, 1 << 15])
def test_unfold_fold_pad_vs_reference_loops(rh, oracle, n):
    from oracle import compose
    mods = QI60[:3]
    rng = np.random.default_rng(n)
    B = 2
    r_small, r_big = rh.Ring(n, mods), rh.Ring(2 * n, mods)
    ci = _block(rng, mods, B, n)
    std = _block(rng, mods, B, 2 * n)
    # Unfold: receiver = the ring of degree 2n
    p_ci, p_std = rh.DevicePoly.from_numpy(r_small, ci), r_big.NewPoly(B)
    r_big.UnfoldConjugateInvariantToStandard(p_ci, p_std)
    got = p_std.numpy()
    for k in range(B):
        for i in range(3):
            assert np.array_equal(got[k, i], compose.unfold_ci_to_standard(ci[k, i]))
    # Fold: receiver = the ring of degree n; any table of n entries < 2n (here a random one and the identity-like one)
    index = rng.integers(0, 2 * n, size=n, dtype=np.uint64)
    p_s, p_o = rh.DevicePoly.from_numpy(r_big, std), r_small.NewPoly(B)
    r_small.FoldStandardToConjugateInvariant(p_s, index, p_o)
    got = p_o.numpy()
    for k in range(B):
        for i, q in enumerate(mods):
            assert np.array_equal(got[k, i], compose.fold_standard_to_ci(std[k, i], index, q))
    # at a lower level only limbs 0..level exist in the blocks
    v_small, v_big = r_small.AtLevel(1), r_big.AtLevel(1)
    p_s2, p_o2 = rh.DevicePoly.from_numpy(r_big, std[:, :2].copy()), v_small.NewPoly(B)
    v_small.FoldStandardToConjugateInvariant(p_s2, index, p_o2)
    assert np.array_equal(p_o2.numpy(), got[:, :2])
    # Pad: receiver = the ring of degree n, output rows of 2n words whose second half must survive
    small = _block(rng, mods, B, n)
    small[0, :, 0] = 0; small[0, :, 1] = 0; small[1, :, n // 2] = 0          # q - 0 is written as q
    before = _block(rng, mods, B, 2 * n)
    for is_ntt in (True, False):
        p_in, p_out = rh.DevicePoly.from_numpy(r_small, small), rh.DevicePoly.from_numpy(r_big, before)
        r_small.PadDefaultRingToConjugateInvariant(p_in, is_ntt, p_out)
        got = p_out.numpy()
        if n <= 4096:
            for k in range(B):
                for i, q in enumerate(mods):
                    assert np.array_equal(got[k, i], compose.pad_default_to_ci(small[k, i], is_ntt, q, before[k, i])), (is_ntt, k, i)
        else:                                                                 # closed form (pinned to the literal loop at the sizes above)
            h = n // 2
            assert np.array_equal(got[:, :, n:], before[:, :, n:])
            if is_ntt:
                assert np.array_equal(got[:, :, :h], small[:, :, :h]) and np.array_equal(got[:, :, h:n], small[:, :, :h][:, :, ::-1])
            else:
                qs = np.array(mods, dtype=np.uint64)[None, :, None]
                assert (got[:, :, 0] == 0).all() and np.array_equal(got[:, :, 1:h], small[:, :, 1:h])
                assert np.array_equal(got[:, :, h], qs[:, :, 0] - small[:, :, h])
                assert np.array_equal(got[:, :, h + 1:n], (qs - small[:, :, 1:h])[:, :, ::-1])
    # argument checks: degrees that are not n / 2n, in place
    with pytest.raises(rh.RingHipError):
        r_big.UnfoldConjugateInvariantToStandard(p_std, p_std)
    with pytest.raises(rh.RingHipError):
        r_small.FoldStandardToConjugateInvariant(p_s, index[: n // 2], p_o)
    r_small.close(); r_big.close()
