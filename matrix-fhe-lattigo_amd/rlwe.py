"""Host-side mirror of the key-switch callers in core/rlwe: the evaluator methods that
sequence the ring hot path for rotations and relinearisation, on device-resident batches.  Every step is a call into
the HIP library; no key generation, no encoders (those stay with the reference).

  GadgetProduct          core/rlwe/evaluator_gadget_product.go:16-30
  DecomposeNTT           :431-453        GadgetProductHoisted   :326-349
  Automorphism           core/rlwe/evaluator_automorphism.go:14-60      AutomorphismHoisted  :62-105
  ApplyEvaluationKey     core/rlwe/evaluator_evaluationkey.go:37-123 (same ring degree)   Relinearize  :125-153

GadgetProduct covers both branches of GadgetProductLazy (:102-121): gadgetProductMultiplePLazy (levelP >= 1) and
gadgetProductSinglePAndBitDecompLazy (levelP <= 0, optional BaseTwoDecomposition), for NTT- and coefficient-domain ciphertexts.
The hoisted forms and the automorphism / relinearisation callers take NTT-domain ciphertexts and levelP >= 1."""
import numpy as np

from .ringhip import BasisExtender, DevicePoly, RingHipError, _check, lib
from .schemes import Ciphertext


class GadgetCiphertext:
    """rlwe.GadgetCiphertext (core/rlwe/gadgetciphertext.go:17-45) in the layout the kernels read:
    Q part [digit][component < 2][limb of ringQ][N], P part [digit][component][limb of ringP][N], NTT domain, Montgomery
    form, shared by every ciphertext of a batch."""

    def __init__(self, ringQ, ringP, valueQ, valueP, BaseTwoDecomposition=0, digits_per_limb=None):
        """valueQ / valueP: (rows, 2, limbs, N).  rows = RNS digits; with a power-of-two decomposition (BaseTwoDecomposition > 0,
        at most one P modulus) row sum(digits_per_limb[:i]) + j holds Value[i][j] and digits_per_limb[i] = len(Value[i]).
        ringP / valueP may be None: a gadget ciphertext without P (LevelP() == -1)."""
        valueQ = np.asarray(valueQ, dtype=np.uint64)
        if valueQ.ndim != 4 or valueQ.shape[1] != 2 or valueQ.shape[2] != ringQ.L:
            raise RingHipError("GadgetCiphertext: expected a (rows, 2, limbs of ringQ, N) array for Q")
        self.digits = valueQ.shape[0]
        self.Q = DevicePoly.from_numpy(ringQ, valueQ.reshape(self.digits * 2, ringQ.L, ringQ.N))
        self.P = None
        if ringP is not None:
            valueP = np.asarray(valueP, dtype=np.uint64)
            if valueP.ndim != 4 or valueP.shape[:2] != valueQ.shape[:2] or valueP.shape[2] != ringP.L:
                raise RingHipError("GadgetCiphertext: expected a (rows, 2, limbs of ringP, N) array for P")
            self.P = DevicePoly.from_numpy(ringP, valueP.reshape(self.digits * 2, ringP.L, ringP.N))
        self.levelQ, self.levelP = ringQ.L - 1, (ringP.L - 1 if ringP is not None else -1)
        self.BaseTwoDecomposition = int(BaseTwoDecomposition)
        self.digits_per_limb = list(digits_per_limb) if digits_per_limb is not None else None
        if self.BaseTwoDecomposition and (self.levelP > 0 or self.digits_per_limb is None or sum(self.digits_per_limb) != self.digits):
            raise RingHipError("GadgetCiphertext: BaseTwoDecomposition needs at most one P modulus and digits_per_limb summing to the row count")

    def LevelQ(self):
        return self.levelQ

    def LevelP(self):
        return self.levelP


class PolyQP:
    """ringqp.Poly: the Q part and the P part of one polynomial of the extended ring (ring/ringqp/poly.go)"""

    def __init__(self, Q, P):
        self.Q, self.P = Q, P


class ElementQP:
    """Element[ringqp.Poly] as the lazy key-switch routines use it: Value[0..1] of PolyQP batches and the NTT flag"""

    def __init__(self, value, is_ntt=True):
        self.Value = list(value)
        self.IsNTT = bool(is_ntt)

    @classmethod
    def alloc(cls, ringQ, ringP, npoly, levelQ, levelP):
        mk = lambda: PolyQP(DevicePoly(ringQ.AtLevel(levelQ), npoly, levelQ + 1), DevicePoly(ringP.AtLevel(levelP), npoly, levelP + 1))
        return cls([mk(), mk()], True)

    def LevelP(self):
        return self.Value[0].P.limbs - 1

    def LevelQ(self):
        return self.Value[0].Q.limbs - 1


class Evaluator:
    """rlwe.Evaluator restricted to the key-switch path; `galois_keys` maps a Galois element to its GadgetCiphertext."""

    def __init__(self, ringQ, ringP=None, galois_keys=None):
        self.ringQ, self.ringP = ringQ, ringP
        self.be = BasisExtender(ringQ, ringP)
        self.galois_keys = dict(galois_keys or {})
        self._pool = {}

    def close(self):
        self._pool.clear()
        self.be.close()

    def buffer(self, tag, ring, npoly, limbs):
        """evaluator-owned scratch poly (the reference's eval.BuffQP / BuffCt): allocated once per shape, reused by every call --
        no hipMalloc / hipFree (and their implicit synchronisation) on the hot path.  Not thread-safe, like the reference's
        buffers (use one Evaluator per thread: Evaluator.ShallowCopy in the reference)."""
        key = (tag, id(ring._h), npoly, limbs)
        p = self._pool.get(key)
        if p is None:
            p = self._pool[key] = DevicePoly(ring, npoly, limbs)
        return p

    @staticmethod
    def _rows(level, *polys):
        """device blocks are strided by level+1 rows per poly (ringhip.h): a batch allocated with more limbs than the level
        it is used at would be read with the wrong stride -- refuse it (a single poly's leading limbs are contiguous)"""
        for p in polys:
            if p is None:
                continue
            if getattr(p, "layout", None) == "block":      # 3N rings: the key-switch kernels and the keys speak the reference's order
                p.ring.ToReferenceOrder(p)
            if p.limbs < level + 1 or (p.limbs != level + 1 and p.npoly > 1):
                raise RingHipError("batch of %d polys with %d limbs used at level %d: allocate it at that level" % (p.npoly, p.limbs, level))

    # ---- core/rlwe/evaluator_gadget_product.go ---------------------------------------------------------------
    def GadgetProduct(self, levelQ, cx, gadgetCt, ct):
        """ct = (<decomp(cx), gadget[0]>, <decomp(cx), gadget[1]>) / P mod Q (:16-30).  ct.IsNTT tells the domain of cx and of the
        result (:14); gadgetCt.LevelP() >= 1 takes gadgetProductMultiplePLazy, <= 0 the single-P / bit-decomposition branch (:109-113)"""
        import ctypes as C
        levelQ = min(levelQ, gadgetCt.LevelQ())
        self._rows(levelQ, cx, ct.Value[0], ct.Value[1])
        L, lp = lib(), gadgetCt.LevelP()
        pP = gadgetCt.P.ptr if gadgetCt.P is not None else None
        if lp >= 1 and ct.IsNTT:
            _check(L.rh_bext_gadget_product(self.be._h, levelQ, lp, cx.ptr, gadgetCt.Q.ptr, pP, gadgetCt.digits, ct.Value[0].ptr, ct.Value[1].ptr, cx.npoly))
        elif lp >= 1:
            _check(L.rh_bext_gadget_product_coeff(self.be._h, levelQ, lp, cx.ptr, gadgetCt.Q.ptr, pP, gadgetCt.digits, ct.Value[0].ptr, ct.Value[1].ptr, cx.npoly))
        else:
            dpl = gadgetCt.digits_per_limb
            arr = (C.c_int * len(dpl))(*dpl) if dpl is not None else None
            _check(L.rh_bext_gadget_product_single_p(self.be._h, levelQ, lp, cx.ptr, 1 if ct.IsNTT else 0, gadgetCt.BaseTwoDecomposition, arr,
                                                     gadgetCt.Q.ptr, pP, gadgetCt.digits, ct.Value[0].ptr, ct.Value[1].ptr, cx.npoly))

    def GadgetProductThenAdd(self, levelQ, cx, gadgetCt, add0, add1, ct):
        """ct[c] = add_c + GadgetProduct(cx)[c] (ring.Add, canonical) -- the Add the callers below issue right after the
        product, folded into ModDown's tile epilogue.  add0 / add1: DevicePoly or None; they may be ct.Value[c] themselves."""
        levelQ = min(levelQ, gadgetCt.LevelQ())
        self._rows(levelQ, cx, add0, add1, ct.Value[0], ct.Value[1])
        _check(lib().rh_bext_gadget_product_then_add(self.be._h, levelQ, gadgetCt.LevelP(), cx.ptr, gadgetCt.Q.ptr, gadgetCt.P.ptr,
                                                     gadgetCt.digits, add0.ptr if add0 is not None else None,
                                                     add1.ptr if add1 is not None else None, ct.Value[0].ptr, ct.Value[1].ptr, cx.npoly))

    def BaseRNSDecompositionVectorSize(self, levelQ, levelP):
        return (levelQ + levelP + 1) // (levelP + 1)                  # core/rlwe/params.go:635-642

    def DecomposeNTT(self, levelQ, levelP, c2, c2IsNTT, decompQP=None):
        """(:431-453) -> (decompQ, decompP): digit i of poly k is row i*npoly + k of each block.  `decompQP` is the
        caller's buffer pair, as in the reference (its BuffDecompQP argument); without it the blocks are allocated here."""
        beta = self.BaseRNSDecompositionVectorSize(levelQ, levelP)
        rq, rp = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        self._rows(levelQ, c2)
        if decompQP is None:
            decompQP = (DevicePoly(rq, beta * c2.npoly, levelQ + 1), DevicePoly(rp, beta * c2.npoly, levelP + 1))
        dq, dp = decompQP
        if (dq.npoly, dq.limbs, dp.npoly, dp.limbs) != (beta * c2.npoly, levelQ + 1, beta * c2.npoly, levelP + 1):
            raise RingHipError("DecomposeNTT: decompQP must hold %d polys of %d and %d limbs" % (beta * c2.npoly, levelQ + 1, levelP + 1))
        _check(lib().rh_bext_decompose_ntt(self.be._h, levelQ, levelP, c2.ptr, 1 if c2IsNTT else 0, dq.ptr, dp.ptr, c2.npoly))
        return dq, dp

    def GadgetProductHoisted(self, levelQ, decompQP, gadgetCt, ct):
        """(:326-349) on the output of DecomposeNTT"""
        dq, dp = decompQP
        npoly = ct.Value[0].npoly
        self._rows(levelQ, dq, ct.Value[0], ct.Value[1])
        _check(lib().rh_bext_gadget_product_hoisted(self.be._h, levelQ, gadgetCt.LevelP(), dq.ptr, dp.ptr, gadgetCt.Q.ptr,
                                                    gadgetCt.P.ptr, gadgetCt.digits, ct.Value[0].ptr, ct.Value[1].ptr, npoly))

    def GadgetProductHoistedLazy(self, levelQ, decompQP, gadgetCt, ctQP):
        """(:351-371): the hoisted product WITHOUT the ModDown -- ctQP receives the accumulators modulo Q and modulo P (canonical, NTT
        domain, still scaled by P).  For sums of rotations that share one ModDown (AutomorphismHoistedLazy, linear transformations)."""
        dq, dp = decompQP
        q0, q1, p0, p1 = ctQP.Value[0].Q, ctQP.Value[1].Q, ctQP.Value[0].P, ctQP.Value[1].P
        levelP = gadgetCt.LevelP()
        if ctQP.LevelP() < levelP:
            raise RingHipError("ctQP.LevelP()=%d < gadgetCt.LevelP()=%d" % (ctQP.LevelP(), levelP))
        self._rows(levelQ, dq, q0, q1)
        self._rows(levelP, dp, p0, p1)          # the P accumulators are strided by levelP+1 rows per poly too
        _check(lib().rh_bext_gadget_product_hoisted_lazy(self.be._h, levelQ, levelP, dq.ptr, dp.ptr, gadgetCt.Q.ptr, gadgetCt.P.ptr,
                                                         gadgetCt.digits, q0.ptr, q1.ptr, p0.ptr, p1.ptr, q0.npoly))
        ctQP.IsNTT = True

    def GadgetProductLazy(self, levelQ, cx, gadgetCt, ctQP, cxIsNTT=True):
        """(:100-120) for gadget ciphertexts with more than one P modulus: the product without the ModDown.  The accumulators are the
        canonical residues the hoisted form leaves (gadgetProductMultiplePLazy and ...Hoisted differ in WHEN the digits are formed, not
        in what is summed), so this is DecomposeNTT into the evaluator's buffer + GadgetProductHoistedLazy."""
        levelP = gadgetCt.LevelP()
        dec = (self.buffer("lazydecQ", self.ringQ.AtLevel(levelQ), self.BaseRNSDecompositionVectorSize(levelQ, levelP) * cx.npoly, levelQ + 1),
               self.buffer("lazydecP", self.ringP.AtLevel(levelP), self.BaseRNSDecompositionVectorSize(levelQ, levelP) * cx.npoly, levelP + 1))
        self.DecomposeNTT(levelQ, levelP, cx, cxIsNTT, dec)
        self.GadgetProductHoistedLazy(levelQ, dec, gadgetCt, ctQP)

    def DecomposeSingleNTT(self, levelQ, levelP, nbPi, decompRNS, c2NTT, c2InvNTT, c2QiQ, c2QiP):
        """(:455-478): digit `decompRNS` of the decomposition -- DecomposeAndSplit of the coefficient-domain c2InvNTT, the digit's own limbs
        copied from the NTT-domain c2NTT, the others transformed (the reference's NTTLazy there; canonical here, same residues)"""
        rq, rp = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        self.be.DecomposeAndSplit(levelQ, levelP, nbPi, decompRNS, c2InvNTT, c2QiQ, c2QiP)
        rq.NTT(c2QiQ, c2QiQ)
        rp.NTT(c2QiP, c2QiP)
        st = decompRNS * nbPi
        ed = min(st + nbPi, levelQ + 1)
        N = rq.N
        for k in range(c2NTT.npoly):                   # limbs [st, ed) of every poly from the NTT-domain input (:467-468)
            off = (k * (levelQ + 1) + st) * N * 8
            src = DevicePoly(rq.AtLevel(ed - st - 1), 1, ed - st, ptr=c2NTT.ptr + off, owner=c2NTT)
            dst = DevicePoly(rq.AtLevel(ed - st - 1), 1, ed - st, ptr=c2QiQ.ptr + off, owner=c2QiQ)
            rq.AtLevel(ed - st - 1).CopyLvl(src, dst)

    def ModDown(self, levelQ, levelP, ctQP, ct):
        """(:33-98), NTT -> NTT and coefficient -> coefficient: ct_c = ModDownQPtoQ(NTT)(ctQP_c.Q, ctQP_c.P)"""
        if ctQP.IsNTT != ct.IsNTT:
            raise RingHipError("ModDown: the mixed-domain forms are not built on the device path")
        if ctQP.IsNTT:
            self._rows(levelQ, ct.Value[0], ct.Value[1], ctQP.Value[0].Q, ctQP.Value[1].Q)
            self._rows(levelP, ctQP.Value[0].P, ctQP.Value[1].P)
            _check(lib().rh_bext_moddown_qp_to_q_ntt_pair(self.be._h, levelQ, levelP, ctQP.Value[0].Q.ptr, ctQP.Value[1].Q.ptr,
                                                          ctQP.Value[0].P.ptr, ctQP.Value[1].P.ptr, ct.Value[0].ptr, ct.Value[1].ptr,
                                                          ct.Value[0].npoly))
        else:
            for c in (0, 1):
                self.be.ModDownQPtoQ(levelQ, levelP, ctQP.Value[c].Q, ctQP.Value[c].P, ct.Value[c])

    def ALlocateDecompositionBuffer(self, levelQ, levelP, npoly):
        """(:480-494), spelled as in the reference: the (decompQ, decompP) pair DecomposeNTT fills"""
        beta = self.BaseRNSDecompositionVectorSize(levelQ, levelP)
        return (DevicePoly(self.ringQ.AtLevel(levelQ), beta * npoly, levelQ + 1), DevicePoly(self.ringP.AtLevel(levelP), beta * npoly, levelP + 1))

    # ---- core/rlwe/evaluator_evaluationkey.go ---------------------------------------------------------------
    def ApplyEvaluationKey(self, ctIn, evk, opOut):
        """(:37-123), same ring degree on both sides (:97-99 -> applyEvaluationKey :105-112):
        opOut = (ctIn[0] + KS(ctIn[1])_0, KS(ctIn[1])_1).  The ring-degree switching branches are not built."""
        if ctIn.Degree() != 1 or opOut.Degree() != 1:
            raise RingHipError("cannot ApplyEvaluationKey: input and output Ciphertext must be of degree 1")
        if not ctIn.IsNTT:
            raise RingHipError("ApplyEvaluationKey: coefficient-domain ciphertexts are not supported by the device path")
        level = min(ctIn.Level(), opOut.Level())
        ringQ = self.ringQ.AtLevel(level)
        npoly = ctIn.Value[1].npoly
        self.GadgetProductThenAdd(level, ctIn.Value[1], evk, ctIn.Value[0], None, opOut)   # (:108-111) with the Add in the epilogue
        opOut.IsNTT = True

    def Relinearize(self, ctIn, opOut, rlk=None):
        """(:125-153): degree 2 -> degree 1 with the relinearisation key (galois_keys["rlk"] or the argument)"""
        if ctIn.Degree() != 2:
            raise RingHipError("cannot relinearize: ctIn.Degree() should be 2 but is %d" % ctIn.Degree())
        rlk = rlk or self.galois_keys.get("rlk")
        if rlk is None:
            raise RingHipError("cannot relinearize: relinearization key is missing")
        if not ctIn.IsNTT:
            raise RingHipError("Relinearize: coefficient-domain ciphertexts are not supported by the device path")
        level = min(ctIn.Level(), opOut.Level())
        ringQ = self.ringQ.AtLevel(level)
        npoly = ctIn.Value[2].npoly
        self.GadgetProductThenAdd(level, ctIn.Value[2], rlk, ctIn.Value[0], ctIn.Value[1], opOut)   # (:144-146)
        opOut.IsNTT = True

    # ---- core/rlwe/evaluator_automorphism.go -----------------------------------------------------------------
    def _galois_key(self, galEl):
        if galEl not in self.galois_keys:                             # CheckAndGetGaloisKey
            raise RingHipError("cannot apply Automorphism: GaloisKey[%d] is missing" % galEl)
        return self.galois_keys[galEl]

    def _check_degree1_ntt(self, ctIn, opOut, who):
        if ctIn.Degree() != 1 or opOut.Degree() != 1:
            raise RingHipError("cannot apply %s: input and output Ciphertext must be of degree 1" % who)
        if not ctIn.IsNTT:
            raise RingHipError("%s: coefficient-domain ciphertexts are not supported by the device path" % who)

    def Automorphism(self, ctIn, galEl, opOut):
        """(:14-60): opOut = phi_galEl(ctIn[0] + KS(ctIn[1])_0, KS(ctIn[1])_1)"""
        self._check_degree1_ntt(ctIn, opOut, "Automorphism")
        level = min(ctIn.Level(), opOut.Level())
        ringQ = self.ringQ.AtLevel(level)
        if galEl == 1:
            if opOut is not ctIn:
                for a, b in zip(ctIn.Value, opOut.Value):
                    ringQ.vec_op("ADD_SCALAR_LAZY", a, None, b, s0=[0] * (level + 1))    # opOut.Copy(ctIn): x + 0, no reduction
            opOut.IsNTT = ctIn.IsNTT
            return
        evk = self._galois_key(galEl)
        npoly = ctIn.Value[1].npoly
        tmp = Ciphertext([self.buffer("auto0", ringQ, npoly, level + 1), self.buffer("auto1", ringQ, npoly, level + 1)], is_ntt=True)
        self.GadgetProductThenAdd(level, ctIn.Value[1], evk, ctIn.Value[0], None, tmp)         # product + ringQ.Add (:42-44)
        ringQ.AutomorphismNTT(tmp.Value[0], galEl, opOut.Value[0])   # AutomorphismNTTWithIndex (ring/automorphism.go:52-73)
        ringQ.AutomorphismNTT(tmp.Value[1], galEl, opOut.Value[1])
        opOut.IsNTT = ctIn.IsNTT

    def AutomorphismHoisted(self, level, ctIn, c1DecompQP, galEl, opOut):
        """(:62-105): as Automorphism with the decomposition of ctIn[1] shared between rotations"""
        self._check_degree1_ntt(ctIn, opOut, "AutomorphismHoisted")
        ringQ = self.ringQ.AtLevel(level)
        if galEl == 1:
            return self.Automorphism(ctIn, 1, opOut)
        evk = self._galois_key(galEl)
        npoly = ctIn.Value[1].npoly
        tmp = Ciphertext([self.buffer("auto0", ringQ, npoly, level + 1), self.buffer("auto1", ringQ, npoly, level + 1)], is_ntt=True)
        dq, dp = c1DecompQP
        self._rows(level, dq, ctIn.Value[0], opOut.Value[0], opOut.Value[1])
        _check(lib().rh_bext_gadget_product_hoisted_then_add(self.be._h, level, evk.LevelP(), dq.ptr, dp.ptr, evk.Q.ptr, evk.P.ptr, evk.digits,
                                                             ctIn.Value[0].ptr, None, tmp.Value[0].ptr, tmp.Value[1].ptr, npoly))   # product + Add (:88-89)
        ringQ.AutomorphismNTT(tmp.Value[0], galEl, opOut.Value[0])
        ringQ.AutomorphismNTT(tmp.Value[1], galEl, opOut.Value[1])
        opOut.IsNTT = ctIn.IsNTT

    def AutomorphismHoistedLazy(self, levelQ, ctIn, c1DecompQP, galEl, ctQP):
        """(:103-160), NTT-domain ctQP: the rotated ciphertext modulo QP and scaled by P --
        ctQP[1] = phi(KS_1), ctQP[0] = phi(KS_0 + P * ctIn[0]) on the Q part, phi(KS_0) on the P part (P * ctIn[0] vanishes modulo P)"""
        evk = self._galois_key(galEl)
        levelP = evk.LevelP()
        if ctQP.LevelP() < levelP:
            raise RingHipError("ctQP.LevelP()=%d < GaloisKey[%d].LevelP()=%d" % (ctQP.LevelP(), galEl, levelP))
        if not ctIn.IsNTT:
            raise RingHipError("AutomorphismHoistedLazy: coefficient-domain ciphertexts are not supported by the device path")
        ringQ, ringP = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        npoly = ctIn.Value[0].npoly
        tmp = ElementQP([PolyQP(self.buffer("lazyQ0", ringQ, npoly, levelQ + 1), self.buffer("lazyP0", ringP, npoly, levelP + 1)),
                         PolyQP(self.buffer("lazyQ1", ringQ, npoly, levelQ + 1), self.buffer("lazyP1", ringP, npoly, levelP + 1))])
        self.GadgetProductHoistedLazy(levelQ, c1DecompQP, evk, tmp)
        # "Result NTT domain is returned according to the NTT flag of ctQP" (:105): the flag only selects which index map is applied
        # (:134-157) -- ringQP.AutomorphismNTTWithIndex or the coefficient-domain ringQP.Automorphism -- and is left as the caller set it
        autQ, autP = (ringQ.AutomorphismNTT, ringP.AutomorphismNTT) if ctQP.IsNTT else (ringQ.Automorphism, ringP.Automorphism)
        autQ(tmp.Value[1].Q, galEl, ctQP.Value[1].Q)                                      # ringQP.Automorphism(NTTWithIndex) (:136 / :147)
        autP(tmp.Value[1].P, galEl, ctQP.Value[1].P)
        P = 1
        for p in self.ringP.moduli[:levelP + 1]:
            P *= int(p)
        ringQ.MulScalarBigintThenAdd(ctIn.Value[0], P, tmp.Value[0].Q)                    # + ctIn[0] * P (:138-142 as one pass: same canonical values)
        autQ(tmp.Value[0].Q, galEl, ctQP.Value[0].Q)                                      # (:144 / :155)
        autP(tmp.Value[0].P, galEl, ctQP.Value[0].P)
