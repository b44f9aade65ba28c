"""Oracle COMPOSITIONS: the reference's call sequences for the callers of the ring hot path, restated over the pinned oracle
pieces (ring_oracle.c).  TEST INFRASTRUCTURE ONLY (tests/, bench.py's verification leg)."""
import os
import re

import numpy as np

from . import ring_oracle as orc

_ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OPS = {m.group(1): int(m.group(2)) for m in re.finditer(r"RH_OP_([A-Z0-9_]+)\s*=\s*(\d+)", open(os.path.join(_ROOT, "include", "ringhip_ops.h")).read())}


def gadget_product(N, Q, P, levelQ, levelP, cx, evkQ, evkP):
    """rlwe.Evaluator.GadgetProduct, NTT-domain input, levelP >= 1 (core/rlwe/evaluator_gadget_product.go:16-30):
    gadgetProductMultiplePLazy (:123-188) = INTT, per digit DecomposeSingleNTT (:455-478) + MulCoeffsMontgomeryLazy
    (ThenAddLazy) with the periodic Reduce, then ModDown NTT -> NTT (:33-46, ring/basis_extension.go:241-258).
    cx: (levelQ+1, N); evkQ / evkP: (digits, 2, len(Q) / len(P), N).  Returns (ct0, ct1), each (levelQ+1, N)."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], P[:LP]
    srQ = [orc.SubRingConsts(N, q) for q in Ql]
    srP = [orc.SubRingConsts(N, p) for p in Pl]
    beta = (levelQ + levelP + 1) // (levelP + 1)
    cxinv = np.stack([orc.intt(cx[i], srQ[i]) for i in range(LQ)])
    acc = {("Q", 0): None, ("Q", 1): None, ("P", 0): None, ("P", 1): None}
    qiof = int(2.0 ** 64 / float(max(Ql))) >> 1
    piof = int(2.0 ** 64 / float(max(Pl))) >> 1
    reduce = 0

    def red(which, mods):
        for c in (0, 1):
            acc[(which, c)] = np.stack([orc.vec_op(OPS["REDUCE"], acc[(which, c)][i], None, acc[(which, c)][i], 0, 0, mods[i]) for i in range(len(mods))])

    for d in range(beta):
        c2q, c2p = orc.decompose_and_split(levelQ, levelP, LP, d, cxinv, Q, P)
        st, ed = d * LP, min(d * LP + LP, LQ)
        c2q = np.stack([cx[i] if st <= i < ed else orc.ntt(c2q[i], srQ[i]) for i in range(LQ)])
        c2p = np.stack([orc.ntt(c2p[j], srP[j]) for j in range(LP)])
        for c in (0, 1):
            for which, c2, ev, mods in (("Q", c2q, evkQ, Ql), ("P", c2p, evkP, Pl)):
                op = OPS["MUL_MONT_LAZY"] if d == 0 else OPS["MUL_MONT_LAZY_THEN_ADD_LAZY"]
                prev = acc[(which, c)] if d else np.zeros_like(c2)
                acc[(which, c)] = np.stack([orc.vec_op(op, ev[d, c, i], c2[i], prev[i], 0, 0, mods[i]) for i in range(len(mods))])
        if reduce % qiof == qiof - 1:
            red("Q", Ql)
        if reduce % piof == piof - 1:
            red("P", Pl)
        reduce += 1
    if reduce % qiof:
        red("Q", Ql)
    if reduce % piof:
        red("P", Pl)
    return [orc.moddown_qp_to_q_ntt(acc[("Q", c)], acc[("P", c)], Ql, Pl, srQ, srP) for c in (0, 1)]
