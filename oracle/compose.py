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


def _mac(acc, key, c2, mods, first):
    op = OPS["MUL_MONT_LAZY"] if first else OPS["MUL_MONT_LAZY_THEN_ADD_LAZY"]
    prev = acc if not first else np.zeros_like(c2)
    return np.stack([orc.vec_op(op, key[i], c2[i], prev[i], 0, 0, mods[i]) for i in range(len(mods))])


def _reduce(a, mods):
    return np.stack([orc.vec_op(OPS["REDUCE"], a[i], None, a[i], 0, 0, mods[i]) for i in range(len(mods))])


def gadget_product_coeff(N, Q, P, levelQ, levelP, cx, evkQ, evkP):
    """GadgetProduct for a coefficient-domain ciphertext, levelP >= 1: cxNTT = NTT(cx) (:139-143), the lazy product as above,
    ringQP.INTT (:114-118), ModDown INTT -> INTT = ModDownQPtoQ (:62-66).  cx and the results: coefficient domain."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], P[:LP]
    srQ = [orc.SubRingConsts(N, q) for q in Ql]
    srP = [orc.SubRingConsts(N, p) for p in Pl]
    beta = (levelQ + levelP + 1) // (levelP + 1)
    cxntt = np.stack([orc.ntt(cx[i], srQ[i]) for i in range(LQ)])
    accQ, accP = [None, None], [None, None]
    qiof = int(2.0 ** 64 / float(max(Ql))) >> 1
    piof = int(2.0 ** 64 / float(max(Pl))) >> 1
    reduce = 0
    for d in range(beta):
        c2q, c2p = orc.decompose_and_split(levelQ, levelP, LP, d, cx, Q, P)
        st, ed = d * LP, min(d * LP + LP, LQ)
        c2q = np.stack([cxntt[i] if st <= i < ed else orc.ntt(c2q[i], srQ[i]) for i in range(LQ)])
        c2p = np.stack([orc.ntt(c2p[j], srP[j]) for j in range(LP)])
        for c in (0, 1):
            accQ[c] = _mac(accQ[c], evkQ[d, c], c2q, Ql, d == 0)
            accP[c] = _mac(accP[c], evkP[d, c], c2p, Pl, d == 0)
        if reduce % qiof == qiof - 1:
            accQ = [_reduce(a, Ql) for a in accQ]
        if reduce % piof == piof - 1:
            accP = [_reduce(a, Pl) for a in accP]
        reduce += 1
    if reduce % qiof:
        accQ = [_reduce(a, Ql) for a in accQ]
    if reduce % piof:
        accP = [_reduce(a, Pl) for a in accP]
    out = []
    for c in (0, 1):
        q = np.stack([orc.intt(accQ[c][i], srQ[i]) for i in range(LQ)])
        p = np.stack([orc.intt(accP[c][j], srP[j]) for j in range(LP)])
        out.append(orc.moddown_qp_to_q(q, p, Ql, Pl))
    return out


def gadget_product_single_p(N, Q, P, levelQ, levelP, cx, is_ntt, pw2, digits_per_limb, evkQ, evkP):
    """gadgetProductSinglePAndBitDecompLazy (core/rlwe/evaluator_gadget_product.go:190-324) + ModDown (:33-98), levelP in {0, -1}.
    evkQ / evkP: (rows, 2, limbs, N), row = (digits before limb i) + j.  cx / results in the domain `is_ntt` names."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], (P[:LP] if LP else [])
    srQ = [orc.SubRingConsts(N, q) for q in Ql]
    srP = [orc.SubRingConsts(N, p) for p in Pl]
    cxinv = np.stack([orc.intt(cx[i], srQ[i]) for i in range(LQ)]) if is_ntt else np.asarray(cx, dtype=np.uint64)
    mask = (1 << pw2) - 1 if pw2 else 0
    qiof = int(2.0 ** 64 / float(max(Ql))) >> 1
    piof = (int(2.0 ** 64 / float(max(Pl))) >> 1) if LP else 1
    accQ, accP = [None, None], [None, None]
    reduce = e = 0
    for i in range(LQ):
        nd = digits_per_limb[i] if pw2 else 1
        if not pw2:
            c2q, c2p = orc.decompose_and_split(levelQ, levelP, levelP + 1, i, cxinv, Q, P)
        for j in range(nd):
            if pw2:
                cw = orc.vec_op(OPS["MASK"], cxinv[i], None, np.zeros(N, dtype=np.uint64), j * pw2, mask, Ql[i])
                c2q = np.stack([cw for _ in range(LQ)])
                c2p = np.stack([cw for _ in range(LP)]) if LP else None
            nq = np.stack([orc.ntt(c2q[u], srQ[u], lazy=True) for u in range(LQ)])                    # s.NTTLazy (:258-262)
            for c in (0, 1):
                accQ[c] = _mac(accQ[c], evkQ[e, c], nq, Ql, e == 0)
            if LP:
                npp = np.stack([orc.ntt(c2p[u], srP[u], lazy=True) for u in range(LP)])
                for c in (0, 1):
                    accP[c] = _mac(accP[c], evkP[e, c], npp, Pl, e == 0)
            if reduce % qiof == qiof - 1:
                accQ = [_reduce(a, Ql) for a in accQ]
            if LP and reduce % piof == piof - 1:
                accP = [_reduce(a, Pl) for a in accP]
            reduce += 1
            e += 1
    if reduce % qiof:
        accQ = [_reduce(a, Ql) for a in accQ]
    if LP and reduce % piof:
        accP = [_reduce(a, Pl) for a in accP]
    out = []
    for c in (0, 1):
        if is_ntt:
            out.append(orc.moddown_qp_to_q_ntt(accQ[c], accP[c], Ql, Pl, srQ, srP) if LP else accQ[c])
        else:
            q = np.stack([orc.intt(accQ[c][i], srQ[i]) for i in range(LQ)])
            if LP:
                p = np.stack([orc.intt(accP[c][j], srP[j]) for j in range(LP)])
                q = orc.moddown_qp_to_q(q, p, Ql, Pl)
            out.append(q)
    return out


# ---- standard <-> conjugate-invariant bridges (ring/conjugate_invariant.go), literal restatements of the reference's loops -------------
def unfold_ci_to_standard(ci):
    """UnfoldConjugateInvariantToStandard (:8-26) on one limb: copy, then tmp2[jdx] = tmp1[idx] for idx = N-1 .. 0, jdx = N .. 2N-1"""
    ci = np.asarray(ci, dtype=np.uint64)
    N = ci.size
    out = np.empty(2 * N, dtype=np.uint64)
    out[:N] = ci
    idx, jdx = N - 1, N
    while jdx < 2 * N:
        out[jdx] = ci[idx]
        idx, jdx = idx - 1, jdx + 1
    return out


def fold_standard_to_ci(std, index, q):
    """FoldStandardToConjugateInvariant (:31-49) on one limb: AutomorphismNTTWithIndex over the N outputs (ring/automorphism.go:50-78),
    then SubRing.Add (addvec: CRed(x + y), ring/vec_ops.go:7-29) with the first N words of the standard poly"""
    std = np.asarray(std, dtype=np.uint64)
    N = std.size // 2
    out = std[np.asarray(index[:N], dtype=np.int64)].copy()
    return orc.vec_op(OPS["ADD"], out, std[:N].copy(), out, 0, 0, q)


def pad_default_to_ci(std, is_ntt, q, ci_before):
    """PadDefaultRingToConjugateInvariant (:52-80) on one limb, the in-place loop run literally (its second half reads what its first
    half wrote); ci_before: the 2N words the output limb held (words N..2N-1 are not written)"""
    std = np.asarray(std, dtype=np.uint64)
    N = std.size
    tmp = [int(x) for x in ci_before]
    tmp[:N] = [int(x) for x in std]
    if is_ntt:
        for j in range(N):
            tmp[N - j - 1] = tmp[j]
    else:
        tmp[0] = 0
        for j in range(1, N):
            tmp[N - j] = int(q) - tmp[j]
    return np.array(tmp, dtype=np.uint64)


def external_product(N, Q, P, levelQ, levelP, ct, is_ntt, rgswQ, rgswP):
    """rgsw.Evaluator.ExternalProduct, LevelP >= 1 (core/rgsw/evaluator.go:42-80 -> externalProductInPlaceMultipleP :188-257 -> ModDownQPtoQNTT):
    ONE pair of lazy accumulators over both components k of the RLWE ciphertext and every digit i, the Reduce counter running through
    (:229-241), the closing Reduce (:245-253).  ct: (2, levelQ+1, N) in the NTT domain (is_ntt) or the coefficient domain (:208-216);
    rgswQ / rgswP: (2, digits, 2, len(Q) / len(P), N) = rgsw.Value[k].Value[i][0][c].{Q, P}.  Returns (out0, out1), NTT domain."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], P[:LP]
    srQ = [orc.SubRingConsts(N, q) for q in Ql]
    srP = [orc.SubRingConsts(N, p) for p in Pl]
    beta = (levelQ + levelP + 1) // (levelP + 1)
    qiof = int(2.0 ** 64 / float(max(Ql))) >> 1
    piof = int(2.0 ** 64 / float(max(Pl))) >> 1
    acc = {}
    reduce = 0
    for k in (0, 1):
        if is_ntt:
            c2ntt = ct[k]
            c2inv = np.stack([orc.intt(ct[k][i], srQ[i]) for i in range(LQ)])
        else:
            c2inv = ct[k]
            c2ntt = np.stack([orc.ntt(ct[k][i], srQ[i]) for i in range(LQ)])
        for d in range(beta):
            c2q, c2p = orc.decompose_and_split(levelQ, levelP, LP, d, c2inv, Q, P)
            st, ed = d * LP, min(d * LP + LP, LQ)
            c2q = np.stack([c2ntt[i] if st <= i < ed else orc.ntt(c2q[i], srQ[i]) for i in range(LQ)])
            c2p = np.stack([orc.ntt(c2p[j], srP[j]) for j in range(LP)])
            first = k == 0 and d == 0
            for c in (0, 1):
                acc[("Q", c)] = _mac(acc.get(("Q", c)), rgswQ[k][d][c], c2q, Ql, first)
                acc[("P", c)] = _mac(acc.get(("P", c)), rgswP[k][d][c], c2p, Pl, first)
            if reduce % qiof == qiof - 1:
                for c in (0, 1):
                    acc[("Q", c)] = _reduce(acc[("Q", c)], Ql)
            if reduce % piof == piof - 1:
                for c in (0, 1):
                    acc[("P", c)] = _reduce(acc[("P", c)], Pl)
            reduce += 1
    if reduce % qiof:
        for c in (0, 1):
            acc[("Q", c)] = _reduce(acc[("Q", c)], Ql)
    if reduce % piof:
        for c in (0, 1):
            acc[("P", c)] = _reduce(acc[("P", c)], Pl)
    return [orc.moddown_qp_to_q_ntt(acc[("Q", c)], acc[("P", c)], Ql, Pl, srQ, srP) for c in (0, 1)]


def external_product_single_p(N, Q, P, levelQ, levelP, ct, pw2, digits_per_limb, rgswQ, rgswP):
    """rgsw.Evaluator.ExternalProduct for LevelP in {0, -1} (core/rgsw/evaluator.go:55-70): externalProductInPlaceSinglePAndBitDecomp
    (:119-186) then ModDownQPtoQNTT (LevelP = 0) or CopyLvl.  ct: (2, levelQ+1, N), NTT domain; rgswQ / rgswP: per component k a
    (rows, 2, limbs, N) array, row = (digits before limb i) + j  (rgswP: None without P).  mask = all ones when pw2 = 0 (:134-137);
    every product is the canonical MulCoeffsMontgomery(ThenAdd) (:155-178)."""
    LQ, LP = levelQ + 1, levelP + 1
    Ql, Pl = Q[:LQ], (P[:LP] if LP else [])
    srQ = [orc.SubRingConsts(N, q) for q in Ql]
    srP = [orc.SubRingConsts(N, p) for p in Pl]
    mask = (1 << pw2) - 1 if pw2 else 0xFFFFFFFFFFFFFFFF
    accQ, accP = [None, None], [None, None]
    first = True
    for k in (0, 1):
        cinv = np.stack([orc.intt(ct[k][i], srQ[i]) for i in range(LQ)])
        e = 0
        for i in range(LQ):
            for j in range(digits_per_limb[i] if pw2 else 1):
                cw = orc.vec_op(OPS["MASK"], cinv[i], None, np.zeros(N, dtype=np.uint64), j * pw2, mask, Ql[i])
                op = OPS["MUL_MONT"] if first else OPS["MUL_MONT_THEN_ADD"]
                for mods, srs, key, acc in ((Ql, srQ, rgswQ[k], accQ), (Pl, srP, rgswP[k] if LP else None, accP)):
                    if not mods:
                        continue
                    cwn = [orc.ntt(cw, srs[u], lazy=True) for u in range(len(mods))]              # s.NTTLazy(cw, cwNTT)
                    for c in (0, 1):
                        prev = acc[c] if not first else np.zeros((len(mods), N), dtype=np.uint64)
                        acc[c] = np.stack([orc.vec_op(op, key[e, c, u], cwn[u], prev[u], 0, 0, mods[u]) for u in range(len(mods))])
                first = False
                e += 1
    if LP:
        return [orc.moddown_qp_to_q_ntt(accQ[c], accP[c], Ql, Pl, srQ, srP) for c in (0, 1)]
    return accQ
