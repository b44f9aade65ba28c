"""Host-side mirror of core/rgsw's external product on device batches (a caller of the key-switch path, SURVEY 8(b) "Callers").

  Ciphertext               core/rgsw/elements.go:15-25   (two gadget ciphertexts: Value[0], Value[1])
  Evaluator.ExternalProduct   core/rgsw/evaluator.go:42-80, externalProductInPlaceMultipleP :188-257 (LevelP >= 1)

RLWE (c0, c1) x RGSW = (<decomp(c0), RGSW[0]> + <decomp(c1), RGSW[1]>) / P: two gadget products summed modulo QP BEFORE the one ModDown.
The reference accumulates both products in one pair of lazy accumulators with a running Reduce counter and closes with a Reduce; every
Reduce is the canonical residue, so the accumulators after the loop are the canonical residues of the sum whatever the schedule -- here
the two lazy products (rlwe.Evaluator.GadgetProductLazy: canonical residues modulo Q and modulo P) are added with ring.Add and handed to
ModDown: the same bits (tests/test_gpu_rgsw.py checks it against the reference's loop restated over the oracle pieces).  LevelP <= 0 takes
externalProductInPlaceSinglePAndBitDecomp (:119-186) the same way; the 32-bit branch (:82-117, one modulus below 2^29) is not built."""
from .ringhip import RingHipError
from .rlwe import ElementQP, Evaluator as RLWEEvaluator, GadgetCiphertext


class Ciphertext:
    """rgsw.Ciphertext: Value[0] encrypts (P*w*m, 0)-gadget of the message against c0, Value[1] against c1 (core/rgsw/elements.go:15-25)"""

    def __init__(self, value0, value1):
        if not isinstance(value0, GadgetCiphertext) or not isinstance(value1, GadgetCiphertext):
            raise RingHipError("rgsw.Ciphertext: two rlwe.GadgetCiphertext values")
        if (value0.LevelQ(), value0.LevelP()) != (value1.LevelQ(), value1.LevelP()):
            raise RingHipError("rgsw.Ciphertext: the two gadget ciphertexts must share their levels")
        self.Value = [value0, value1]

    def LevelQ(self):
        return self.Value[0].LevelQ()

    def LevelP(self):
        return self.Value[0].LevelP()


class Evaluator(RLWEEvaluator):
    """rgsw.Evaluator (core/rgsw/evaluator.go:12-35): an rlwe.Evaluator plus the external product"""

    def ExternalProduct(self, op0, op1, opOut):
        """opOut = op0 x op1 (:42-80).  op0, opOut: degree-1 rlwe ciphertexts (batches), NTT or coefficient domain like the reference
        (:208-216: the decomposition is taken from whichever domain op0 is in); op1: rgsw.Ciphertext with LevelP >= 1.  The product runs
        at op1's levels (:44); opOut may be op0."""
        levelQ, levelP = op1.LevelQ(), op1.LevelP()
        if levelP < 1:
            return self._external_product_single_p(op0, op1, opOut, levelQ, levelP)
        if len(op0.Value) != 2 or len(opOut.Value) != 2:
            raise RingHipError("ExternalProduct: degree-1 ciphertexts")
        if opOut.IsNTT is not True:
            raise RingHipError("ExternalProduct: the result is in the NTT domain (ModDownQPtoQNTT, :75-76)")
        self._rows(levelQ, op0.Value[0], op0.Value[1], opOut.Value[0], opOut.Value[1])
        npoly = op0.Value[0].npoly
        acc = [ElementQP.alloc(self.ringQ, self.ringP, npoly, levelQ, levelP) for _ in (0, 1)]
        for k in (0, 1):
            self.GadgetProductLazy(levelQ, op0.Value[k], op1.Value[k], acc[k], cxIsNTT=op0.IsNTT)
        rq, rp = self.ringQ.AtLevel(levelQ), self.ringP.AtLevel(levelP)
        for c in (0, 1):
            rq.Add(acc[0].Value[c].Q, acc[1].Value[c].Q, acc[0].Value[c].Q)
            rp.Add(acc[0].Value[c].P, acc[1].Value[c].P, acc[0].Value[c].P)
        self.ModDown(levelQ, levelP, acc[0], opOut)

    def _external_product_single_p(self, op0, op1, opOut, levelQ, levelP):
        """LevelP <= 0 (:55-70): externalProductInPlaceSinglePAndBitDecomp (:119-186) -- every digit is MaskVec of a limb of INTT(c_k) under
        every modulus (mask = all ones without a power-of-two decomposition), the products accumulated with the canonical
        MulCoeffsMontgomery(ThenAdd) over both components k -- then ModDownQPtoQNTT (LevelP = 0) or CopyLvl (no P).  Each component's sum is
        rh_bext_gadget_product_single_p_lazy in its rgsw digit form; the two canonical sums are added with ring.Add.  The reference's 32-bit
        branch (:54-57, :82-117: one modulus below 2^29, plain wrapping products) is not built and refused."""
        import ctypes as C
        from .ringhip import _check, lib, DevicePoly
        if levelQ == 0 and levelP == -1 and (int(self.ringQ.moduli[0]) >> 29) == 0:
            raise RingHipError("ExternalProduct: the 32-bit branch (core/rgsw/evaluator.go:82-117) is not built")
        if not op0.IsNTT or opOut.IsNTT is not True:
            raise RingHipError("ExternalProduct (LevelP <= 0): NTT-domain ciphertexts (the reference takes INTT of op0.Value[k], :148)")
        self._rows(levelQ, op0.Value[0], op0.Value[1], opOut.Value[0], opOut.Value[1])
        npoly = op0.Value[0].npoly
        rq = self.ringQ.AtLevel(levelQ)
        rp = self.ringP.AtLevel(levelP) if levelP >= 0 else None
        accQ = [[DevicePoly(rq, npoly, levelQ + 1) for _ in (0, 1)] for _ in (0, 1)]
        accP = [[DevicePoly(rp, npoly, levelP + 1) for _ in (0, 1)] for _ in (0, 1)] if rp is not None else None
        for k in (0, 1):
            g = op1.Value[k]
            dpl = g.digits_per_limb
            arr = (C.c_int * len(dpl))(*dpl) if dpl is not None else None
            _check(lib().rh_bext_gadget_product_single_p_lazy(
                self.be._h, levelQ, levelP, op0.Value[k].ptr, 1, g.BaseTwoDecomposition, arr, g.Q.ptr, g.P.ptr if g.P is not None else None,
                g.digits, 1, accQ[k][0].ptr, accQ[k][1].ptr, accP[k][0].ptr if accP else None, accP[k][1].ptr if accP else None, npoly))
        for c in (0, 1):
            rq.Add(accQ[0][c], accQ[1][c], accQ[0][c])
            if rp is not None:
                rp.Add(accP[0][c], accP[1][c], accP[0][c])
        if rp is not None:
            _check(lib().rh_bext_moddown_qp_to_q_ntt_pair(self.be._h, levelQ, levelP, accQ[0][0].ptr, accQ[0][1].ptr, accP[0][0].ptr, accP[0][1].ptr,
                                                          opOut.Value[0].ptr, opOut.Value[1].ptr, npoly))
        else:
            for c in (0, 1):
                rq.CopyLvl(accQ[0][c], opOut.Value[c])
