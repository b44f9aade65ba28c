"""GPU: RNS rescale (ring/scaling.go) through the C ABI, bit-exact against the oracle and big integers."""
import numpy as np
import pytest

from conftest import QI60
from test_oracle_bext import prod, rns
from test_oracle_rescale import div_round

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("round_", [0, 1])
@pytest.mark.parametrize("N,L,nb", [(64, 4, 3), (4096, 6, 1), (8192, 6, 5), (64, 2, 1), (1 << 15, 8, 2), (8192, 5, 1), (1 << 16, 4, 1), (1 << 14, 2, 1)])
def test_div_many_coefficient_and_ntt_domain(rh, oracle, N, L, nb, round_):
    Q = QI60[:L]
    ring = rh.Ring(N, Q)
    rng = np.random.default_rng(N + L + nb + round_)
    big = prod(Q)
    B = 2
    vals = [[int.from_bytes(rng.bytes(big.bit_length() // 8 + 2), "little") % big // 10 for _ in range(N)] for _ in range(B)]
    a = np.stack([rns(v, Q) for v in vals])
    exp = np.stack([oracle.div_by_last_modulus_many(a[k], Q, nb, round_) for k in range(B)])
    # coefficient domain, output block with level+1-nb limbs
    p0 = rh.DevicePoly.from_numpy(ring, a)
    p1 = rh.DevicePoly(ring, B, L - nb)
    (ring.DivRoundByLastModulusMany if round_ else ring.DivFloorByLastModulusMany)(nb, p0, p1)
    got = p1.numpy()
    assert np.array_equal(got, exp)
    for i in range(L - nb):                                   # and against exact integers (ring_test.go:242-331)
        want = list(vals[1][:32])
        for j in range(nb):
            m = Q[L - 1 - j]
            want = [div_round(v, m) if round_ else v // m for v in want]
        assert [int(x) for x in got[1, i][:32]] == [w % Q[i] for w in want]
    # NTT domain: NTT(input) -> Div...ManyNTT -> equals NTT of the coefficient-domain result; output block keeps L limbs
    pn = rh.DevicePoly.from_numpy(ring, a)
    ring.NTT(pn, pn)
    before = pn.numpy()
    po = rh.DevicePoly.from_numpy(ring, np.zeros((B, L, N), dtype=np.uint64))
    (ring.DivRoundByLastModulusManyNTT if round_ else ring.DivFloorByLastModulusManyNTT)(nb, pn, po)
    assert np.array_equal(pn.numpy(), before)                 # input untouched
    sub = ring.AtLevel(L - nb - 1)
    chk = rh.DevicePoly.from_numpy(sub, po.numpy()[:, :L - nb].copy())
    sub.INTT(chk, chk)
    assert np.array_equal(chk.numpy(), exp)
    assert not po.numpy()[:, L - nb:].any()                   # limbs above the new level are not written
    ring.close()


def _ntt_prime_below(bits, two_n, skip=0):
    """largest primes p < 2^bits with p = 1 mod 2N (deterministic Miller-Rabin for 64-bit integers)"""
    def is_prime(n):
        if n < 2:
            return False
        for p in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            if n % p == 0:
                return n == p
        d, s = n - 1, 0
        while d % 2 == 0:
            d //= 2; s += 1
        for a in (2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37):
            x = pow(a, d, n)
            if x in (1, n - 1):
                continue
            for _ in range(s - 1):
                x = x * x % n
                if x == n - 1:
                    break
            else:
                return False
        return True
    p = ((1 << bits) // two_n) * two_n + 1
    while True:
        p -= two_n
        if is_prime(p):
            if skip == 0:
                return p
            skip -= 1


@pytest.mark.parametrize("round_", [0, 1])
def test_rescale_with_moduli_of_different_sizes(rh, oracle, round_):
    # last modulus 61 bits, the others 50 bits: qL + q > 8q, so the re-expansion must be reduced modulo each limb's q (the
    # hand-scheduled column stages that skip this reduction are not eligible; the launcher falls back)
    N = 1 << 14
    Q = [_ntt_prime_below(50, 2 * N, 0), _ntt_prime_below(50, 2 * N, 1), QI60[0]]
    ring = rh.Ring(N, Q)
    rng = np.random.default_rng(50 + round_)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q]) for _ in range(2)])
    exp = np.stack([oracle.div_by_last_modulus_many(a[k], Q, 1, round_) for k in range(2)])
    pn = rh.DevicePoly.from_numpy(ring, a)
    ring.NTT(pn, pn)
    po = rh.DevicePoly.from_numpy(ring, np.zeros((2, 3, N), dtype=np.uint64))
    (ring.DivRoundByLastModulusManyNTT if round_ else ring.DivFloorByLastModulusManyNTT)(1, pn, po)
    sub = ring.AtLevel(1)
    chk = rh.DevicePoly.from_numpy(sub, po.numpy()[:, :2].copy())
    sub.INTT(chk, chk)
    assert np.array_equal(chk.numpy(), exp)
    ring.close()


@pytest.mark.parametrize("kind,N,block_order", [("3n", 3 << 6, 0), ("3n", 3 << 13, 0), ("3n", 3 << 14, 1), ("ci", 1 << 14, 0), ("ci", 256, 0)])
@pytest.mark.parametrize("nb", [1, 2])
def test_rescale_on_3n_and_conjugate_invariant_rings(rh, oracle, kind, N, block_order, nb):
    # schemes/matrix_ckks/evaluator.go:235 rescales on the 3N ring (ring.DivRoundByLastModulusManyNTT through the ring's own transform); the
    # same for the conjugate-invariant ring.  NTT(input) -> Div...ManyNTT -> INTT == the coefficient-domain division (RNS per coefficient,
    # the ring type does not enter), which is pinned against the oracle.
    from test_gpu_schemes import primes_3n
    L, B = 4, 2
    if kind == "3n":
        mods = primes_3n(oracle, N, L)
        ring = rh.Ring(N, mods, kind=rh.Matrix3N)
        if block_order:
            ring.set_tuning("ntt3n_block_order", 1)
    else:
        mods = QI60[:L]
        ring = rh.Ring(N, mods, kind=rh.ConjugateInvariant)
    rng = np.random.default_rng(N + nb)
    a = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in mods]) for _ in range(B)])
    for round_ in (0, 1):
        exp = np.stack([oracle.div_by_last_modulus_many(a[k], mods, nb, round_) for k in range(B)])
        pc = rh.DevicePoly.from_numpy(ring, a)                    # coefficient domain
        p1 = rh.DevicePoly(ring, B, L - nb)
        (ring.DivRoundByLastModulusMany if round_ else ring.DivFloorByLastModulusMany)(nb, pc, p1)
        assert np.array_equal(p1.numpy(), exp)
        pn = rh.DevicePoly.from_numpy(ring, a)
        ring.NTT(pn, pn)
        po = rh.DevicePoly(ring, B, L - nb)
        (ring.DivRoundByLastModulusManyNTT if round_ else ring.DivFloorByLastModulusManyNTT)(nb, pn, po)
        sub = ring.AtLevel(L - nb - 1)
        sub.INTT(po, po)
        assert np.array_equal(po.numpy(), exp)
    ring.close()
