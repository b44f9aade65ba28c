"""Scheme-level callers of the ring hot path for BASELINE configs 3 and 4: the *sequences* of ring calls the reference's
evaluators issue, on device-resident batches.  Nothing here computes: every step is a Ring method (HIP kernels behind
the C ABI).  No key material, encoders or encryptors -- those stay with the reference.

* ``MatrixCKKSEvaluator.Mul``  schemes/matrix_ckks/evaluator.go:114-192   (config 4, 3N ring)
* ``ckks_tensor_degree1``      schemes/ckks/evaluator.go:821-834          (config 3 / first half of MulRelin)
"""
from .ringhip import RingHipError


class Ciphertext:
    """rlwe.Ciphertext as far as the evaluator reads it: Value[0..degree] (DevicePoly batches of equal shape),
    IsNTT and the level (= limbs - 1).  A batch of B ciphertexts is one Ciphertext whose polys have npoly = B."""

    def __init__(self, value, is_ntt=False):
        self.Value = list(value)
        self.IsNTT = bool(is_ntt)

    def Degree(self):
        return len(self.Value) - 1

    def Level(self):
        return self.Value[0].limbs - 1


class MatrixCKKSEvaluator:
    """schemes/matrix_ckks/evaluator.go: Evaluator over the 3N ring Z_Q[X]/(X^N - X^{N/2} + 1)."""

    def __init__(self, ringQ, block_order=True):
        """block_order (default): device-resident NTT-domain polys of the 3N ring are kept in BLOCK order (Ring.ntt3n_layout = "block"): a
        transform is 2 HBM passes instead of 3, and since every NTT-domain operation of a ring is coefficient-wise, Mul / Add / Rescale chains
        give bit-identical results.  The layout is a TAG on each device block (DevicePoly.layout), not a mode of the data a caller can see:
        Ring.NTT tags what it writes, INTT and the coefficient-wise calls read the tags, a block-order operand that meets a reference-order
        one (e.g. NTT-domain data uploaded from the host) is converted first, and DevicePoly.numpy() always hands out the reference's order
        (ring/ntt_3n.go:82-109).  Rings too small for it (N < 3 * 2^13, or 3^b with b > 1) and block_order=False use the reference's order."""
        self.ringQ = ringQ
        self.fused_tensor = True
        from .ringhip import lib
        if block_order and lib().rh_ring_ntt3n_block_order_supported(ringQ._h):
            ringQ.ntt3n_layout = "block"
        elif not block_order:
            ringQ.ntt3n_layout = None

    def Mul(self, ct0, ct1, ctOut):
        """evaluator.go:114-192.  Reproduced as written, including its two side effects: inputs not yet in the NTT
        domain are transformed IN PLACE and their IsNTT flips (:136-149), and the products are MulCoeffsMontgomery
        on operands that were never put in Montgomery form (no MForm), so every output coefficient carries a factor
        2^-64 mod q_i.  The output is returned in the coefficient domain (:182-189)."""
        if ct0.Level() != ct1.Level():
            raise RingHipError("ciphertexts must be at the same level for multiplication")
        d0, d1 = ct0.Degree(), ct1.Degree()
        if d0 > 1 or d1 > 1:
            raise RingHipError("unsupported ciphertext degrees for multiplication: %d, %d" % (d0, d1))
        if ctOut.Degree() != d0 + d1:
            raise RingHipError("ctOut must have degree %d" % (d0 + d1))
        rq = self.ringQ.AtLevel(ct0.Level())
        for ct in (ct0, ct1):
            if not ct.IsNTT:
                for v in ct.Value:
                    rq.NTT(v, v)
                ct.IsNTT = True
        a, b, o = ct0.Value, ct1.Value, ctOut.Value
        if d0 == 0 and d1 == 0:
            rq.MulCoeffsMontgomery(a[0], b[0], o[0])
        elif d0 == 0 and d1 == 1:
            rq.MulCoeffsMontgomery(a[0], b[0], o[0])
            rq.MulCoeffsMontgomery(a[0], b[1], o[1])
        elif d0 == 1 and d1 == 0:
            rq.MulCoeffsMontgomery(a[0], b[0], o[0])
            rq.MulCoeffsMontgomery(a[1], b[0], o[1])
        elif self.fused_tensor:                      # the four ring calls below as one kernel (same bits)
            rq.TensorDegree1(a[0], a[1], b[0], b[1], o[0], o[1], o[2], mform_first=False)
        else:
            rq.MulCoeffsMontgomery(a[0], b[0], o[0])
            rq.MulCoeffsMontgomery(a[0], b[1], o[1])
            rq.MulCoeffsMontgomeryThenAdd(a[1], b[0], o[1])
            rq.MulCoeffsMontgomery(a[1], b[1], o[2])
        for v in o:
            rq.INTT(v, v)
        ctOut.IsNTT = False


    # ---- MulByConst (evaluator.go:322-380): a ciphertext times a constant -------------------------------------------------------
    @staticmethod
    def _round_to_prec(x, prec):
        """big.Float.SetPrec(prec).SetFloat64 / SetInt: the value rounded to `prec` significant bits, round-half-even (exact when it fits)"""
        from fractions import Fraction
        x = Fraction(x)
        if x == 0:
            return x
        n, e = abs(x), 0
        while n >= (1 << prec):
            n /= 2; e += 1
        while n < (1 << (prec - 1)):
            n *= 2; e -= 1
        fl = n.numerator // n.denominator
        rem = n - fl
        if rem > Fraction(1, 2) or (rem == Fraction(1, 2) and fl & 1):
            fl += 1
        v = Fraction(fl) * (Fraction(2) ** e)
        return v if x > 0 else -v

    @staticmethod
    def _scaled_int(x, scale):
        """bigComplexToRNSScalar (:382-411): x * scale, plus / minus 0.5 by sign, truncated toward zero (exact: the 128-bit scale precision
        holds the 53 + 61-bit product of a float64 constant and a modulus)"""
        from fractions import Fraction
        v = Fraction(x) * scale
        if x > 0:
            v += Fraction(1, 2)
        elif x < 0:
            v -= Fraction(1, 2)
        return int(v)                                   # Python truncates toward zero like big.Float.Int

    def MulByConst(self, ct, constant, ctOut, encoding_precision=53, roots_forward_1=None):
        """evaluator.go:322-380: ctOut = ct * constant.  Integer constants multiply as they are (scale 1); other constants are scaled by the
        moduli one rescaling consumes (q_level ...), rounded half away from zero, and the product's scale grows by that factor (returned; scale
        bookkeeping is the reference Ciphertext's metadata).  The RNS scalars are formed on the host exactly as the reference forms them and
        applied with ring.MulDoubleRNSScalar (the first N/2 coefficients times one scalar, the rest times the other).  A constant with an
        imaginary part mixes in SubRing.RootsForward[1] of the FULL ring (:361-365): hand it over per limb (`roots_forward_1`, Montgomery form,
        as the Go side holds it) -- this mirror does not regenerate the tables a Matrix ring carries for the wrong order."""
        from fractions import Fraction
        if ct.Level() != ctOut.Level():
            raise RingHipError("ciphertexts must be at the same level for constant multiplication")
        if ct.Degree() != ctOut.Degree():
            raise RingHipError("MulByConst: ctOut must have the degree of ct")
        level = ct.Level()
        rq = self.ringQ.AtLevel(level)
        mods = [int(q) for q in self.ringQ.moduli[:level + 1]]
        re_, im_ = (constant.real, constant.imag) if isinstance(constant, complex) else (constant, 0)
        rnd = lambda x: Fraction(x) if isinstance(x, int) else self._round_to_prec(x, encoding_precision)
        re_, im_ = rnd(re_), rnd(im_)
        if re_.denominator == 1 and im_.denominator == 1:          # cmplxBig.IsInt(): no scaling required
            scale = 1
        else:
            scale = mods[level]
            for i in range(1, self.levelsConsumedPerRescaling()):
                scale *= mods[level - i]
        real, imag = self._scaled_int(re_, scale), self._scaled_int(im_, scale)
        rns_real = [real % q for q in mods]                         # NewRNSScalarFromBigint: the non-negative residue
        rns_imag = [imag % q for q in mods]
        if imag != 0:
            if roots_forward_1 is None:
                raise RingHipError("MulByConst: a constant with an imaginary part needs SubRing.RootsForward[1] of every limb (roots_forward_1)")
            rinv = [pow(1 << 64, -1, q) for q in mods]
            rns_imag = [(x * int(w) * ri) % q for x, w, ri, q in zip(rns_imag, roots_forward_1, rinv, mods)]       # ring.MRed(RNSImag[i], RootsForward[1])
        s0 = [(a + b) % q for a, b, q in zip(rns_real, rns_imag, mods)]                                            # CRed(real + imag)
        s1 = [(a + q - b) % q for a, b, q in zip(rns_real, rns_imag, mods)]                                        # CRed(real + q - imag)
        for vin, vout in zip(ct.Value, ctOut.Value):
            rq.MulDoubleRNSScalar(vin, s0, s1, vout)
        ctOut.IsNTT = ct.IsNTT
        return scale

    def levelsConsumedPerRescaling(self):
        """evaluator.go:313-319"""
        return 1

    def Rescale(self, op0, opOut):
        """evaluator.go:208-243: divides every component by the last modulus of its level with ring.DivRoundByLastModulusManyNTT on the
        3N ring (`nbRescales` = levelsConsumedPerRescaling() times); opOut's components hold op0.Level() + 1 - nbRescales limbs.  Like the
        reference it calls the NTT-domain form whatever the ciphertext's IsNTT says, and copies the flag.  (Scale bookkeeping is host-side
        metadata of the reference's Ciphertext and not part of this mirror.)"""
        nb = self.levelsConsumedPerRescaling()
        if op0.Level() <= nb - 1:
            raise RingHipError("cannot Rescale: input Ciphertext level is too low")
        if op0.Degree() != opOut.Degree():
            raise RingHipError("cannot Rescale: opOut must have the degree of op0")
        rq = self.ringQ.AtLevel(op0.Level())
        for vin, vout in zip(op0.Value, opOut.Value):
            rq.DivRoundByLastModulusManyNTT(nb, vin, vout)
        opOut.IsNTT = op0.IsNTT

    def Add(self, ct0, ct1, ctOut):
        """evaluator.go:60-102: component-wise ring.Add over the common degree, the longer ciphertext's remaining components copied
        (CopyLvl); ctOut holds max(degree) + 1 components and takes ct0's domain flag"""
        if ct0.Level() != ct1.Level():
            raise RingHipError("ciphertexts must be at the same level for addition")
        hi, lo = max(ct0.Degree(), ct1.Degree()), min(ct0.Degree(), ct1.Degree())
        if ctOut.Degree() != hi:
            raise RingHipError("ctOut must have degree %d" % hi)
        rq = self.ringQ.AtLevel(ct0.Level())
        for i in range(lo + 1):
            rq.Add(ct0.Value[i], ct1.Value[i], ctOut.Value[i])
        longer = ct0 if ct0.Degree() > ct1.Degree() else ct1
        for i in range(lo + 1, hi + 1):
            rq.CopyLvl(longer.Value[i], ctOut.Value[i])
        ctOut.IsNTT = ct0.IsNTT


    def ModDown(self, op0, opOut, levels):
        """evaluator.go:259-290: drops `levels` limbs WITHOUT dividing (the leading limbs are copied); opOut's components hold
        op0.Level() + 1 - levels limbs"""
        if op0.Level() <= levels - 1:
            raise RingHipError("cannot ModDown: input Ciphertext level is too low")
        if opOut.Degree() != op0.Degree() or opOut.Level() != op0.Level() - levels:
            raise RingHipError("ModDown: opOut must have op0's degree and level %d" % (op0.Level() - levels))
        rq = self.ringQ.AtLevel(opOut.Level())
        for vin, vout in zip(op0.Value, opOut.Value):
            rq.CopyLvl(vin, vout)
        opOut.IsNTT = op0.IsNTT

    def _new(self, degree, level, npoly, is_ntt=False):
        from .ringhip import DevicePoly
        return Ciphertext([DevicePoly(self.ringQ.AtLevel(level), npoly, level + 1) for _ in range(degree + 1)], is_ntt)

    def AddNew(self, ct0, ct1):
        """:104-111"""
        out = self._new(max(ct0.Degree(), ct1.Degree()), ct0.Level(), ct0.Value[0].npoly)
        self.Add(ct0, ct1, out)
        return out

    def MulNew(self, ct0, ct1):
        """:195-200"""
        out = self._new(ct0.Degree() + ct1.Degree(), ct0.Level(), ct0.Value[0].npoly)
        self.Mul(ct0, ct1, out)
        return out

    def RescaleNew(self, ct):
        """:246-251"""
        out = self._new(ct.Degree(), ct.Level() - self.levelsConsumedPerRescaling(), ct.Value[0].npoly)
        self.Rescale(ct, out)
        return out

    def ModDownNew(self, ct, levels):
        """:293-297"""
        out = self._new(ct.Degree(), ct.Level() - levels, ct.Value[0].npoly)
        self.ModDown(ct, out, levels)
        return out

    def DropLevelNew(self, op0, levels):
        """:307-311 (DropLevel :301-303 resizes in place: on device blocks that is a copy of the leading limbs)"""
        return self.ModDownNew(op0, levels)


def ckks_tensor_degree1(ringQ, ct0, ct1, c0, c1, c2, c00, c01):
    """schemes/ckks/evaluator.go:821-834 (degree-1 x degree-1 tensoring of mulRelin; all operands in the NTT domain):
    c00 = MForm(ct0[0]); c01 = MForm(ct0[1]); c0 = c00*ct1[0]; c1 = c00*ct1[1] + c01*ct1[0]; c2 = c01*ct1[1]."""
    ringQ.MForm(ct0.Value[0], c00)
    ringQ.MForm(ct0.Value[1], c01)
    ringQ.MulCoeffsMontgomery(c00, ct1.Value[0], c0)
    ringQ.MulCoeffsMontgomery(c00, ct1.Value[1], c1)
    ringQ.MulCoeffsMontgomeryThenAdd(c01, ct1.Value[0], c1)
    ringQ.MulCoeffsMontgomery(c01, ct1.Value[1], c2)


def ckks_polymul(ringQ, a, b, c, tmp=None, fused=True):
    """BASELINE config 3: c = INTT(NTT(a) . NTT(b)) with MForm + MulCoeffsMontgomery as mulRelin sequences it
    (schemes/ckks/evaluator.go:821-834).  a and b are transformed in place (they end in the NTT domain).
    fused (default): MForm, MulCoeffsMontgomery and the inverse transform as ONE call (Ring.INTTMul: the product is formed on
    load by the inverse transform's first kernel) -- same canonical values; fused=False: the five ring calls as written;
    fused="tile": Ring.PolyMul -- the forward tile stages of both operands, the product and the inverse tile stages as one kernel
    (a and b are CONSUMED: they do not end as NTT(a), NTT(b); same c)."""
    if fused == "tile":
        ringQ.PolyMul(a, b, c)
        return
    if fused:
        ringQ.NTTMany([(a, a), (b, b)])
    else:
        ringQ.NTT(a, a)
        ringQ.NTT(b, b)
    if fused:
        ringQ.INTTMul(a, b, c)
        return
    ringQ.MForm(a, tmp)
    ringQ.MulCoeffsMontgomery(tmp, b, c)
    ringQ.INTT(c, c)
