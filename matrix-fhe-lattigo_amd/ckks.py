"""Host-side mirror of the CKKS evaluator methods that traverse the ring hot path: the ct x ct multiply
with relinearisation and the rescale that follows it, on device-resident batches in the NTT domain.  Call sequences
only -- the arithmetic is the HIP library's; scales, encoders and key generation stay with the reference.

  mulRelin   schemes/ckks/evaluator.go:786-881      Rescale   schemes/ckks/evaluator.go:500-535

A batch of B ciphertexts is one Ciphertext whose polys have npoly = B.  All operands of one call sit at the same level
(every poly block has level+1 limbs): the device layout is (poly, limb, N) contiguous, so a poly cannot be read at a
lower level than it was allocated for -- drop limbs first (DropLevel = re-upload / view of fewer limbs)."""
from .ringhip import DevicePoly, RingHipError
from .schemes import Ciphertext
from . import rlwe


class Evaluator:
    def __init__(self, ringQ, ringP=None, rlk=None, levels_consumed_per_rescaling=1):
        self.ringQ, self.ringP, self.rlk = ringQ, ringP, rlk
        self.nb_rescales = int(levels_consumed_per_rescaling)
        self.fused_tensor = True          # regular ct x ct case: rh_ring_tensor_degree1 instead of six element-wise launches (same bits)
        self.ks = rlwe.Evaluator(ringQ, ringP) if ringP is not None else None
        self._pool = {}

    def close(self):
        self._pool.clear()
        if self.ks:
            self.ks.close()

    def _buffer(self, tag, ring, npoly, limbs):
        key = (tag, id(ring._h), npoly, limbs)
        p = self._pool.get(key)
        if p is None:
            p = self._pool[key] = DevicePoly(ring, npoly, limbs)
        return p

    def _same_level(self, *cts):
        lv = {c.Level() for c in cts}
        if len(lv) != 1:
            raise RingHipError("operands must sit at the same level, got %s" % sorted(lv))
        return lv.pop()

    def MulRelin(self, op0, op1, opOut, relin=True):
        """mulRelin (:786-881).  Degree-1 x degree-1: tensoring (:821-834, squaring case :825-829 when op1 is op0), then
        with relin the gadget product of c2 with the relinearisation key and two Adds (:836-852); opOut has degree 1 with
        relin, 2 without.  Degree-0 x degree-1 (plaintext x ciphertext): MForm + MulCoeffsMontgomery per component (:855-878)."""
        if not (op0.IsNTT and op1.IsNTT):
            raise RingHipError("MulRelin: operands must be in the NTT domain")
        level = self._same_level(op0, op1, opOut)
        rq = self.ringQ.AtLevel(level)
        npoly = op0.Value[0].npoly
        counter = [0]

        def new():                         # eval.buffQ[i]: evaluator-owned, allocated once per shape
            counter[0] += 1
            return self._buffer("buffQ%d" % counter[0], rq, npoly, level + 1)
        d0, d1 = op0.Degree(), op1.Degree()
        if d0 == 1 and d1 == 1:
            need = 1 if relin else 2
            if opOut.Degree() != need:
                raise RingHipError("MulRelin: opOut must have degree %d" % need)
            c00, c01 = new(), new()
            c0, c1 = opOut.Value[0], opOut.Value[1]
            c2 = new() if relin else opOut.Value[2]
            tmp0, tmp1 = (op1, op0) if op1 is opOut else (op0, op1)        # avoid overwriting when the second input is the output
            if op0 is op1 or not self.fused_tensor:
                rq.MForm(tmp0.Value[0], c00)
                rq.MForm(tmp0.Value[1], c01)
            if op0 is op1:                                                  # squaring
                rq.MulCoeffsMontgomery(c00, tmp1.Value[0], c0)
                rq.MulCoeffsMontgomery(c01, tmp1.Value[1], c2)
                rq.MulCoeffsMontgomery(c00, tmp1.Value[1], c1)
                rq.Add(c1, c1, c1)
            elif self.fused_tensor:                                         # the same six ring calls as one kernel
                rq.TensorDegree1(tmp0.Value[0], tmp0.Value[1], tmp1.Value[0], tmp1.Value[1], c0, c1, c2)
            else:
                rq.MulCoeffsMontgomery(c00, tmp1.Value[0], c0)
                rq.MulCoeffsMontgomery(c01, tmp1.Value[1], c2)
                rq.MulCoeffsMontgomery(c00, tmp1.Value[1], c1)
                rq.MulCoeffsMontgomeryThenAdd(c01, tmp1.Value[0], c1)
            if relin:
                if self.rlk is None or self.ks is None:
                    raise RingHipError("cannot MulRelin: Relinearize: relinearization key is missing")
                self.ks.GadgetProductThenAdd(level, c2, self.rlk, c0, c1, opOut)      # GadgetProduct + the two Adds (:850-852)
        elif d0 + d1 == 1 or (d0 == 0 and d1 == 0):
            pt, ct = (op0, op1) if d0 == 0 else (op1, op0)
            if opOut.Degree() != max(d0, d1):
                raise RingHipError("MulRelin: opOut must have degree %d" % max(d0, d1))
            c0 = new()
            rq.MForm(pt.Value[0], c0)
            for i, v in enumerate(ct.Value):
                rq.MulCoeffsMontgomery(c0, v, opOut.Value[i])
        else:
            raise RingHipError("MulRelin: unsupported degrees %d, %d" % (d0, d1))
        opOut.IsNTT = True

    def Rescale(self, op0, opOut):
        """Rescale (:500-535): DivRoundByLastModulusManyNTT(nbRescales) on every component.  opOut's polys keep op0's limb
        count; limbs 0 .. level-nbRescales hold the result (ring/scaling.go:130-156)."""
        nb = self.nb_rescales
        if op0.Level() <= nb - 1:
            raise RingHipError("cannot Rescale: input Ciphertext level is too low")
        if opOut.Degree() != op0.Degree():
            raise RingHipError("Rescale: degrees differ")
        rq = self.ringQ.AtLevel(op0.Level())
        for a, b in zip(op0.Value, opOut.Value):
            rq.DivRoundByLastModulusManyNTT(nb, a, b)
        opOut.IsNTT = op0.IsNTT
