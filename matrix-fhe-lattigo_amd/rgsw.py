"""Host-side mirror of core/rgsw's external product on device batches (a caller of the key-switch path, SURVEY 8(b) "Callers").

  Ciphertext               core/rgsw/elements.go:15-25   (two gadget ciphertexts: Value[0], Value[1])
  Evaluator.ExternalProduct   core/rgsw/evaluator.go:42-80, externalProductInPlaceMultipleP :188-257 (LevelP >= 1)

RLWE (c0, c1) x RGSW = (<decomp(c0), RGSW[0]> + <decomp(c1), RGSW[1]>) / P: two gadget products summed modulo QP BEFORE the one ModDown.
The reference accumulates both products in one pair of lazy accumulators with a running Reduce counter and closes with a Reduce; every
Reduce is the canonical residue, so the accumulators after the loop are the canonical residues of the sum whatever the schedule -- here
the two lazy products (rlwe.Evaluator.GadgetProductLazy: canonical residues modulo Q and modulo P) are added with ring.Add and handed to
ModDown: the same bits (tests/test_gpu_rgsw.py checks it against the reference's loop restated over the oracle pieces).  The single-P and
32-bit branches (:82-186) are not built: they serve blind rotations on small rings, off the throughput path."""
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
            raise RingHipError("ExternalProduct: the single-P / 32-bit branches (core/rgsw/evaluator.go:82-186) are not built; LevelP >= 1")
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
