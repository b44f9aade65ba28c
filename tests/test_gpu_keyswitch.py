"""GPU: hybrid key-switch gadget product (core/rlwe/evaluator_gadget_product.go:16-188) against the same sequence composed
from oracle pieces (each pinned elsewhere): INTT -> per digit DecomposeAndSplit, NTT, MulCoeffsMontgomeryLazy(ThenAddLazy)
with periodic Reduce -> ModDownQPtoQNTT.  The evaluation key is uniformly random: arithmetic
parity needs no key generation.  End-to-end key-switch semantics are pinned only statistically in the reference
(core/rlwe/rlwe_test.go:690-798): that part stays "parity unpinned"."""
import numpy as np
import pytest

from conftest import QI60, PI60

pytestmark = pytest.mark.gpu


def oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx, evkQ, evkP):
    """the composition lives in oracle/compose.py (bench.py's verification leg uses it too)"""
    from oracle import compose
    assert compose.OPS == rh.OPS
    return compose.gadget_product(N, Q, P, levelQ, levelP, cx, evkQ, evkP)


@pytest.mark.parametrize("N,nq,np_,levelQ,levelP", [(64, 6, 2, 5, 1), (8192, 8, 3, 7, 2), (4096, 24, 6, 23, 5), (64, 9, 2, 6, 1), (8192, 9, 3, 6, 2), (8192, 8, 3, 7, 1), (4096, 7, 4, 3, 2)])
def test_gadget_product_vs_oracle_composition(rh, oracle, N, nq, np_, levelQ, levelP):
    Q, P = QI60[:nq], PI60[:np_]
    rq, rp = rh.Ring(N, Q), rh.Ring(N, P)
    be = rh.BasisExtender(rq, rp)
    rng = np.random.default_rng(N + nq + levelQ)
    LQ, LP = levelQ + 1, levelP + 1
    beta_key = (nq - 1 + np_) // np_ if levelP == np_ - 1 else (levelQ + levelP + 1) // (levelP + 1)
    beta_key = max(beta_key, (levelQ + levelP + 1) // (levelP + 1))
    npoly = 2
    cx = np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q[:LQ]]) for _ in range(npoly)])
    evkQ = np.stack([np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q) for q in Q]) for _ in range(2)]) for _ in range(beta_key)])
    evkP = np.stack([np.stack([np.stack([rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(p) for p in P]) for _ in range(2)]) for _ in range(beta_key)])
    pcx = rh.DevicePoly.from_numpy(rq.AtLevel(levelQ), cx)
    dq = rh.DevicePoly.from_numpy(rq, evkQ.reshape(beta_key * 2, nq, N))
    dp = rh.DevicePoly.from_numpy(rp, evkP.reshape(beta_key * 2, np_, N))
    ct0, ct1 = rh.DevicePoly(rq, npoly, LQ), rh.DevicePoly(rq, npoly, LQ)
    be.GadgetProduct(levelQ, levelP, pcx, dq.ptr, dp.ptr, beta_key, ct0, ct1)
    g0, g1 = ct0.numpy(), ct1.numpy()
    for k in range(npoly):
        e0, e1 = oracle_gadget_product(oracle, rh, N, Q, P, levelQ, levelP, cx[k], evkQ, evkP)
        assert np.array_equal(g0[k], e0)
        assert np.array_equal(g1[k], e1)
    be.close(); rq.close(); rp.close()
