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
