#!/usr/bin/env python3
"""Secondary measurements (not the headline metric): inverse NTT, element-wise kernels, basis extension and the 3N
transform at the BASELINE config sizes, each against its algorithmic bytes.  Prints one JSON object."""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import torch
import matrix_fhe_lattigo_amd as rh
sys.path.insert(0, os.path.join(ROOT, "tests"))
from conftest import QI60, PI60

dev = torch.device("cuda", 0)
stream = torch.cuda.current_stream()
PEAK = 8000.0


def timed(fn, reps=10, warm=2):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream)
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps


def rand_block(B, mods, N):
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, len(mods), 1)
    return torch.randint(0, 1 << 62, (B, len(mods), N), dtype=torch.int64, device=dev) % qs


def entry(name, ms, alg_bytes, units, unit_name):
    gbs = alg_bytes / (ms * 1e-3) / 1e9
    return {"op": name, "ms": round(ms, 4), unit_name + "_per_s": units / (ms * 1e-3), "algorithmic_GBps": round(gbs, 1), "frac_of_8TBps": round(gbs / PEAK, 3)}


res = []
# ---- N = 2^16, 16 limbs, batch 512 (4 GiB) ----
N, L, B = 1 << 16, 16, 512
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
a, b, c = rand_block(B, QI60[:L], N), rand_block(B, QI60[:L], N), rand_block(B, QI60[:L], N)
pa, pb, pc = (rh.DevicePoly.from_torch(ring, t) for t in (a, b, c))
res.append(entry("Ring.NTT N=2^16 L=16", timed(lambda: ring.NTT(pa, pa)), 16.0 * N * L * B, B, "poly"))
res.append(entry("Ring.INTT N=2^16 L=16", timed(lambda: ring.INTT(pa, pa)), 16.0 * N * L * B, B, "poly"))
res.append(entry("Ring.NTTLazy (exact reference representatives) N=2^16 L=16", timed(lambda: ring.NTTLazy(pa, pc)), 16.0 * N * L * B, B, "poly"))
for op, nops in (("ADD", 3), ("MUL_MONT", 3), ("MUL_MONT_THEN_ADD", 4), ("MFORM", 2), ("MUL_BARRETT", 3), ("REDUCE", 2), ("MUL_MONT_LAZY_THEN_ADD_LAZY", 4)):
    res.append(entry("vec " + op + " N=2^16 L=16", timed(lambda: ring.vec_op(op, pa, pb, pc)), 8.0 * nops * N * L * B, B, "poly"))
# rescale (ring/scaling.go): NTT-domain rounded division by the last modulus, 16 -> 15 limbs
po = rh.DevicePoly.from_torch(ring, torch.empty((B, L - 1, N), dtype=torch.int64, device=dev))
res.append(entry("DivRoundByLastModulusNTT N=2^16 L=16->15 (1 INTT + 15 NTT + 2 elementwise passes)",
                 timed(lambda: ring.DivRoundByLastModulusNTT(pa, po), reps=5), 8.0 * N * (2 * (L - 1) + 1) * B + 16.0 * N * L * B, B, "poly"))
# automorphism in the NTT domain (ring/automorphism.go:12-109): an index gather over every limb
res.append(entry("AutomorphismNTT (X -> X^5) N=2^16 L=16", timed(lambda: ring.AutomorphismNTT(pa, 5, pc)), 16.0 * N * L * B, B, "poly"))
del pa, pb, pc, a, b, c, po
ring.close(); torch.cuda.empty_cache()
# conjugate-invariant ring (ring/ntt.go:716-1311): fold + the negacyclic core on the 4N-th-root tables.  QI60 moduli are 1 mod 2^18 = 4N.
ring = rh.Ring(N, QI60[:L], kind=rh.ConjugateInvariant); ring.set_stream(stream.cuda_stream)
a = rand_block(B, QI60[:L], N); pa = rh.DevicePoly.from_torch(ring, a)
res.append(entry("conjugate-invariant Ring.NTT N=2^16 L=16", timed(lambda: ring.NTT(pa, pa)), 16.0 * N * L * B, B, "poly"))
res.append(entry("conjugate-invariant Ring.INTT N=2^16 L=16", timed(lambda: ring.INTT(pa, pa)), 16.0 * N * L * B, B, "poly"))
del pa, a
ring.close(); torch.cuda.empty_cache()

# ---- config 3: N = 2^15, 16 limbs: c = INTT(NTT(a) * NTT(b)) as schemes/ckks/evaluator.go:821-834 sequences it ----
N, L, B = 1 << 15, 16, 512
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
a, b = rand_block(B, QI60[:L], N), rand_block(B, QI60[:L], N)
pa, pb = rh.DevicePoly.from_torch(ring, a), rh.DevicePoly.from_torch(ring, b)
def polymul():
    ring.NTT(pa, pa); ring.NTT(pb, pb); ring.MForm(pa, pa); ring.MulCoeffsMontgomery(pa, pb, pa); ring.INTT(pa, pa)
ms = timed(polymul, reps=5)
e = entry("config3 poly-mul N=2^15 L=16 (NTT,NTT,MForm,MulCoeffsMontgomery,INTT)", ms, 88.0 * N * L * B, B, "polymul")
e["frac_vs_fused_lower_bound_24NL"] = round(24.0 * N * L * B / (ms * 1e-3) / 1e9 / PEAK, 3)
res.append(e)
def polymul_fused():
    ring.NTT(pa, pa); ring.NTT(pb, pb); ring.INTTMul(pa, pb, pa)
ms = timed(polymul_fused, reps=5)
e = entry("config3 poly-mul N=2^15 L=16, MForm + MulCoeffsMontgomery formed on load by the inverse transform (Ring.INTTMul)", ms, 88.0 * N * L * B, B, "polymul")
e["frac_vs_fused_lower_bound_24NL"] = round(24.0 * N * L * B / (ms * 1e-3) / 1e9 / PEAK, 3)
res.append(e)
def polymul_many():
    ring.NTTMany([(pa, pa), (pb, pb)]); ring.INTTMul(pa, pb, pa)
ms = timed(polymul_many, reps=5)
e = entry("config3 poly-mul N=2^15 L=16, one pipeline through both forward transforms (Ring.NTTMany) + Ring.INTTMul", ms, 88.0 * N * L * B, B, "polymul")
e["frac_vs_fused_lower_bound_24NL"] = round(24.0 * N * L * B / (ms * 1e-3) / 1e9 / PEAK, 3)
res.append(e)
def polymul_tile():
    ring.PolyMul(pa, pb, pa)
ms = timed(polymul_tile, reps=5)
e = entry("config3 poly-mul N=2^15 L=16, Ring.PolyMul: forward tile stages of both operands + product + inverse tile stages as ONE kernel (72 B per coefficient)", ms, 88.0 * N * L * B, B, "polymul")
e["frac_vs_fused_lower_bound_24NL"] = round(24.0 * N * L * B / (ms * 1e-3) / 1e9 / PEAK, 3)
res.append(e)
del pa, pb, a, b
ring.close(); torch.cuda.empty_cache()

# ---- config 5 shapes: N = 2^16, Q = Qi60[0:24], P = Pi60[0:6]: DecomposeAndSplit digit, ModDownQPtoQNTT ----
N, B = 1 << 16, 64
rq, rp = rh.Ring(N, QI60[:24]), rh.Ring(N, PI60[:6]); rq.set_stream(stream.cuda_stream); rp.set_stream(stream.cuda_stream)
be = rh.BasisExtender(rq, rp)
xq, xp = rand_block(B, QI60[:24], N), rand_block(B, PI60[:6], N)
oq, op_ = torch.zeros_like(xq), torch.zeros_like(xp)
pq, pp, poq, pop = rh.DevicePoly.from_torch(rq, xq), rh.DevicePoly.from_torch(rp, xp), rh.DevicePoly.from_torch(rq, oq), rh.DevicePoly.from_torch(rp, op_)
res.append(entry("DecomposeAndSplit N=2^16 alpha=6 -> 18 Q + 6 P limbs", timed(lambda: be.DecomposeAndSplit(23, 5, 6, 1, pq, poq, pop), reps=5), 8.0 * N * (6 + 24) * B, B, "poly"))
res.append(entry("ModUpPtoQ N=2^16 6 -> 24 limbs", timed(lambda: be.ModUpPtoQ(5, 23, pp, poq), reps=5), 8.0 * N * (6 + 24) * B, B, "poly"))
res.append(entry("ModDownQPtoQ N=2^16 (24+6) -> 24 limbs", timed(lambda: be.ModDownQPtoQ(23, 5, pq, pp, poq), reps=5), 8.0 * N * (6 + 24 + 24) * B, B, "poly"))
res.append(entry("ModDownQPtoQNTT N=2^16 (24+6) -> 24 limbs", timed(lambda: be.ModDownQPtoQNTT(23, 5, pq, pp, poq), reps=5), 8.0 * N * (6 + 24 + 24) * B + 16.0 * N * 30 * B, B, "poly"))
# config 5: GadgetProduct (key-switch of one ciphertext component), beta = 4 digits, key shared by the batch
beta = 4
evq, evp = rand_block(beta * 2, QI60[:24], N), rand_block(beta * 2, PI60[:6], N)
c0, c1 = torch.zeros_like(xq), torch.zeros_like(xq)
p0, p1 = rh.DevicePoly.from_torch(rq, c0), rh.DevicePoly.from_torch(rq, c1)
ms = timed(lambda: be.GadgetProduct(23, 5, pq, evq.data_ptr(), evp.data_ptr(), beta, p0, p1), reps=3, warm=1)
limb_ntts = 24 + beta * 24 + 2 * 30                      # INTT(cx) + per digit (18 Q + 6 P) + 2 x ModDownNTT (6 INTT + 24 NTT)
e = entry("config5 GadgetProduct N=2^16 Q=24 P=6 beta=4 (per ciphertext component, key shared by batch of %d)" % B, ms,
          16.0 * N * limb_ntts * B + 2.0 * beta * 30 * 8 * N, B, "keyswitch")
e["limb_ntt_equivalents_per_keyswitch"] = limb_ntts
res.append(e)
# CKKS ct x ct multiply with relinearisation, then rescale (schemes/ckks/evaluator.go:786-881, 500-535) at the config 5 parameters
gct = rh.rlwe.GadgetCiphertext.__new__(rh.rlwe.GadgetCiphertext)
gct.digits, gct.levelQ, gct.levelP = beta, 23, 5
gct.Q, gct.P = rh.DevicePoly.from_torch(rq, evq), rh.DevicePoly.from_torch(rp, evp)
cev = rh.ckks.Evaluator(rq, rp, rlk=gct)
mk = lambda: rh.DevicePoly.from_torch(rq, rand_block(B, QI60[:24], N))
ctA, ctB = rh.Ciphertext([mk(), mk()], is_ntt=True), rh.Ciphertext([mk(), mk()], is_ntt=True)
ctO, ctR = rh.Ciphertext([mk(), mk()], is_ntt=True), rh.Ciphertext([mk(), mk()], is_ntt=True)
def mulrelin_rescale():
    cev.MulRelin(ctA, ctB, ctO, relin=True)
    cev.Rescale(ctO, ctR)
ms = timed(mulrelin_rescale, reps=3, warm=1)
e = entry("CKKS MulRelin + Rescale N=2^16 Q=24 P=6 (tensor, key switch, 2 adds, rescale of both components; batch %d)" % B, ms,
          0.0, B, "ctmul")
e.pop("algorithmic_GBps"); e.pop("frac_of_8TBps")
res.append(e)
# rotations of one ciphertext with a shared decomposition (core/rlwe/evaluator_automorphism.go:62-105): DecomposeNTT once, then per rotation
# GadgetProductHoisted + Add + two NTT-domain automorphisms
gal = 5
kev = rh.rlwe.Evaluator(rq, rp, galois_keys={gal: gct})
dec = kev.DecomposeNTT(23, 5, ctA.Value[1], True)
ms_dec = timed(lambda: kev.DecomposeNTT(23, 5, ctA.Value[1], True, dec), reps=3, warm=1)
ms_rot = timed(lambda: kev.AutomorphismHoisted(23, ctA, dec, gal, ctO), reps=3, warm=1)
e = entry("hoisted rotation N=2^16 Q=24 P=6: DecomposeNTT once (%.3f ms per batch of %d), then per rotation" % (ms_dec, B), ms_rot, 0.0, B, "rotation")
e.pop("algorithmic_GBps"); e.pop("frac_of_8TBps")
res.append(e)
# rotations accumulated modulo QP under ONE ModDown (AutomorphismHoistedLazy, core/rlwe/evaluator_automorphism.go:103-160): the pattern of linear transformations
lz = rh.rlwe.ElementQP.alloc(rq, rp, B, 23, 5)
ms_lazy = timed(lambda: kev.AutomorphismHoistedLazy(23, ctA, dec, gal, lz), reps=3, warm=1)
ms_md = timed(lambda: kev.ModDown(23, 5, lz, ctO), reps=3, warm=1)
e = entry("lazy hoisted rotation N=2^16 Q=24 P=6 (result modulo QP; one ModDown per sum of rotations: %.3f ms per batch of %d)" % (ms_md, B), ms_lazy, 0.0, B, "rotation")
e.pop("algorithmic_GBps"); e.pop("frac_of_8TBps")
res.append(e)
del dec, lz
kev.close()
del ctA, ctB, ctO, ctR, gct
cev.close()
del pq, pp, poq, pop, xq, xp, oq, op_, p0, p1, c0, c1, evq, evp
be.close(); rq.close(); rp.close(); torch.cuda.empty_cache()

# ---- config 2 / 4 rings: 3N transform ----
sys.path.insert(0, os.path.join(ROOT, "tools"))
from primes3n import moduli_3n
for N, L, B in ((3 << 13, 1, 1024), (3 << 16, 24, 16), (3 << 14, 24, 64)):       # config 2; config 4 at both readings of its "logN = 16" (SURVEY 8(d)), the same bytes per batch
    mods = moduli_3n(N, L)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N); ring.set_stream(stream.cuda_stream)
    x = rand_block(B, mods, N)
    px = rh.DevicePoly.from_torch(ring, x)
    res.append(entry("3N NTT N=%d L=%d" % (N, L), timed(lambda: ring.NTT(px, px), reps=5), 16.0 * N * L * B, B, "poly"))
    res.append(entry("3N INTT N=%d L=%d" % (N, L), timed(lambda: ring.INTT(px, px), reps=5), 16.0 * N * L * B, B, "poly"))
    ring.set_tuning("ntt3n_block_order", 1)
    res.append(entry("3N NTT N=%d L=%d, device NTT domain in block order (no permutation pass)" % (N, L), timed(lambda: ring.NTT(px, px), reps=5), 16.0 * N * L * B, B, "poly"))
    res.append(entry("3N INTT N=%d L=%d, block order" % (N, L), timed(lambda: ring.INTT(px, px), reps=5), 16.0 * N * L * B, B, "poly"))
    ring.set_tuning("ntt3n_block_order", 0)
    if L == 24:
        # config 4: matrix_ckks.Evaluator.Mul on degree-1 ciphertexts given in the coefficient domain
        # (schemes/matrix_ckks/evaluator.go:114-192): 4 NTT + 3 MulCoeffsMontgomery + 1 ...ThenAdd + 3 INTT
        blocks = [rh.DevicePoly.from_torch(ring, rand_block(B, mods, N)) for _ in range(7)]
        ct0, ct1 = rh.Ciphertext(blocks[0:2]), rh.Ciphertext(blocks[2:4])
        out = rh.Ciphertext(blocks[4:7])
        ev = rh.MatrixCKKSEvaluator(ring, block_order=False)

        def mul():
            ct0.IsNTT = ct1.IsNTT = False
            ev.Mul(ct0, ct1, out)
        res.append(entry("config4 matrix_ckks Mul N=%d L=%d, reference-order NTT domain (block_order=False)" % (N, L),
                         timed(mul, reps=5), (7 * 16.0 + 3 * 24.0 + 32.0) * N * L * B, B, "ctmul"))
        ev = rh.MatrixCKKSEvaluator(ring)                    # the default since round 3: block order, carried as per-block tags
        res.append(entry("config4 matrix_ckks Mul N=%d L=%d (4 NTT, tensoring, 3 INTT; DEFAULT evaluator: block-order device NTT domain, tagged per block)" % (N, L),
                         timed(mul, reps=5), (7 * 16.0 + 3 * 24.0 + 32.0) * N * L * B, B, "ctmul"))
        ring.ntt3n_layout = None
        del blocks, ct0, ct1, out
    del px, x
    ring.close(); torch.cuda.empty_cache()
print(json.dumps({"device": torch.cuda.get_device_name(0), "results": res}, indent=1))
