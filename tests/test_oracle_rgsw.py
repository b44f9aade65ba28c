"""CPU: the oracle's restatement of the RGSW external product loop (core/rgsw/evaluator.go:188-257) against the gadget-product
composition it generalises: with a zero second gadget ciphertext the external product of (c0, c1) is the gadget product of c0, and the
external product is additive in the RGSW value's two halves (both compositions are built from the same pinned pieces)."""
import numpy as np
import pytest

from conftest import QI60, PI60, uniform_mod
from oracle import compose


def _key(rng, digits, mods, N, zero=False):
    if zero:
        return np.zeros((digits, 2, len(mods), N), dtype=np.uint64)
    return np.stack([np.stack([np.stack([uniform_mod(rng, q, N) for q in mods]) for _ in range(2)]) for _ in range(digits)])


@pytest.mark.parametrize("N,nq,np_,is_ntt", [(64, 4, 2, True), (256, 5, 3, False)])
def test_external_product_reduces_to_the_gadget_product(N, nq, np_, is_ntt):
    import oracle as orc
    Q, P = QI60[:nq], PI60[:np_]
    rng = np.random.default_rng(N + nq)
    levelQ, levelP = nq - 1, np_ - 1
    beta = (levelQ + levelP + 1) // (levelP + 1)
    ct = np.stack([np.stack([uniform_mod(rng, q, N) for q in Q]) for _ in range(2)])
    k0q, k0p = _key(rng, beta, Q, N), _key(rng, beta, P, N)
    zq, zp = _key(rng, beta, Q, N, zero=True), _key(rng, beta, P, N, zero=True)
    e = compose.external_product(N, Q, P, levelQ, levelP, ct, is_ntt, [k0q, zq], [k0p, zp])
    c0_ntt = ct[0] if is_ntt else np.stack([orc.ntt(ct[0][i], orc.SubRingConsts(N, Q[i])) for i in range(nq)])
    g = compose.gadget_product(N, Q, P, levelQ, levelP, c0_ntt, k0q, k0p)
    assert np.array_equal(e[0], g[0]) and np.array_equal(e[1], g[1])
