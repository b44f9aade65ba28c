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
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], P[:LP]
    srQ = [oracle.SubRingConsts(N, q) for q in Q]
    srP = [oracle.SubRingConsts(N, p) for p in P]
    beta = (levelQ + levelP + 1) // (levelP + 1)
    cxinv = np.stack([oracle.intt(cx[i], srQ[i]) for i in range(LQ)])
    OPS = rh.OPS
    acc = {("Q", 0): None, ("Q", 1): None, ("P", 0): None, ("P", 1): None}
    qiof = int(2.0 ** 64 / float(max(Ql))) >> 1
    piof = int(2.0 ** 64 / float(max(Pl))) >> 1
    reduce = 0

    def red(which, mods):
        for c in (0, 1):
            acc[(which, c)] = np.stack([oracle.vec_op(OPS["REDUCE"], acc[(which, c)][i], None, acc[(which, c)][i], 0, 0, mods[i]) for i in range(len(mods))])

    for d in range(beta):
        c2q, c2p = oracle.decompose_and_split(levelQ, levelP, LP, d, cxinv, Q, P)
        st, ed = d * LP, min(d * LP + LP, LQ)
        c2q = np.stack([cx[i] if st <= i < ed else oracle.ntt(c2q[i], srQ[i]) for i in range(LQ)])
        c2p = np.stack([oracle.ntt(c2p[j], srP[j]) for j in range(LP)])
        for c in (0, 1):
            for which, c2, ev, mods in (("Q", c2q, evkQ, Ql), ("P", c2p, evkP, Pl)):
                op = OPS["MUL_MONT_LAZY"] if d == 0 else OPS["MUL_MONT_LAZY_THEN_ADD_LAZY"]
                prev = acc[(which, c)] if d else np.zeros_like(c2)
                acc[(which, c)] = np.stack([oracle.vec_op(op, ev[d, c, i], c2[i], prev[i], 0, 0, mods[i]) for i in range(len(mods))])
        if reduce % qiof == qiof - 1:
            red("Q", Ql)
        if reduce % piof == piof - 1:
            red("P", Pl)
        reduce += 1
    if reduce % qiof:
        red("Q", Ql)
    if reduce % piof:
        red("P", Pl)
    return [oracle.moddown_qp_to_q_ntt(acc[("Q", c)], acc[("P", c)], Ql, Pl, srQ[:LQ], srP[:LP]) for c in (0, 1)]


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
