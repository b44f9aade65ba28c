"""Conjugate-invariant NTT (ring/ntt.go:716-1311).  CPU: the oracle against the reference's own cross-check
(ring/ring_test.go:85-126: squaring in Z[X+X^-1]/(X^2N+1) == squaring the symmetric extension in the standard 2N ring);
GPU: bit-exact against the oracle."""
import numpy as np
import pytest

from conftest import QI60, uniform_mod


def ci_square_via_standard(oracle, a, N, q):
    sr2 = oracle.SubRingConsts(2 * N, q)
    ext = np.zeros(2 * N, dtype=np.uint64)
    ext[:N] = a
    for j in range(1, N):
        ext[2 * N - j] = np.uint64(q) - a[j] if a[j] else 0
    y = oracle.ntt(ext, sr2)
    y = oracle.vec_op(33, y, None, y, 0, 0, q)            # MForm
    y = oracle.vec_op(13, y, y, y, 0, 0, q)               # MulCoeffsMontgomery
    y = oracle.vec_op(35, y, None, y, 0, 0, q)            # IMForm
    return oracle.intt(y, sr2)[:N]


@pytest.mark.parametrize("logN", [4, 5, 8, 10])
def test_oracle_ci_matches_standard_ring(oracle, logN):
    N, q = 1 << logN, QI60[0]
    sr = oracle.SubRingConsts(N, q, nthroot=4 * N)
    rng = np.random.default_rng(logN)
    a = uniform_mod(rng, q, N)
    y = oracle.ntt_ci(a, sr)
    assert np.array_equal(oracle.intt_ci(y, sr), a)
    z = oracle.vec_op(33, y, None, y, 0, 0, q)
    z = oracle.vec_op(13, z, z, z, 0, 0, q)
    z = oracle.vec_op(35, z, None, z, 0, 0, q)
    assert np.array_equal(oracle.intt_ci(z, sr), ci_square_via_standard(oracle, a, N, q))


@pytest.mark.gpu
# logN >= 14: the fold runs inside the hand-scheduled column stages (a thread owns the column pair the fold couples, column 0 apart)
@pytest.mark.parametrize("logN", [4, 8, 12, 13, 14, 15, 16])
def test_gpu_ci_vs_oracle(rh, oracle, logN):
    N, mods = 1 << logN, QI60[:3]
    ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    srs = [oracle.SubRingConsts(N, q, nthroot=4 * N) for q in mods]
    c = ring.constants()
    for i, s in enumerate(srs):                                   # engine-generated 4N-th-root tables == oracle's
        assert np.array_equal(c["roots_fwd"][i], s.roots_fwd) and int(c["ninv"][i]) == s.ninv
    rng = np.random.default_rng(logN)
    a = np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)])
    p, o = rh.DevicePoly.from_numpy(ring, a), ring.NewPoly(2)
    ring.NTT(p, o)
    exp = np.stack([np.stack([oracle.ntt_ci(a[k, i], srs[i]) for i in range(3)]) for k in range(2)])
    assert np.array_equal(o.numpy(), exp)
    ring.NTTLazy(p, o)                                            # lazy contract: congruent, inside [0, 6q-2]
    assert np.array_equal(o.numpy() % np.array(mods, dtype=np.uint64)[None, :, None], exp)
    ring.INTT(o, o)
    assert np.array_equal(o.numpy(), a)
    ring.NTT(p, p)                                                # in place, then back out of place
    assert np.array_equal(p.numpy(), exp)
    ring.INTT(p, o)
    assert np.array_equal(o.numpy(), a) and np.array_equal(p.numpy(), exp)
    ring.set_tuning("fuse_ci", 0)                                 # the fold as a pass of its own: same bits
    ring.INTT(p, p)
    assert np.array_equal(p.numpy(), a)
    ring.NTT(p, p)
    assert np.array_equal(p.numpy(), exp)
    ring.set_tuning("fuse_ci", 1)
    assert np.array_equal(ring.SubRings[1].NTT(a[0, 1]), exp[0, 1])
    assert np.array_equal(ring.SubRings[2].INTT(exp[1, 2]), a[1, 2])
    ring.close()
