#!/usr/bin/env python3
"""One CKKS composite at the config-5 parameters, for rocprofv3 --kernel-trace --stats: bench_ckks.py <mulrelin|rotation|moddown|decompose> [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60, PI60
what = sys.argv[1]; B = int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
N, beta = 1 << 16, 4
rq, rp = rh.Ring(N, QI60[:24]), rh.Ring(N, PI60[:6])
for r in (rq, rp): r.set_stream(stream.cuda_stream)
def rb(n, mods):
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, len(mods), 1)
    return torch.randint(0, 1 << 62, (n, len(mods), N), dtype=torch.int64, device=dev) % qs
evq, evp = rb(2 * beta, QI60[:24]), rb(2 * beta, PI60[:6])
gct = rh.rlwe.GadgetCiphertext.__new__(rh.rlwe.GadgetCiphertext)
gct.digits, gct.levelQ, gct.levelP = beta, 23, 5
gct.Q, gct.P = rh.DevicePoly.from_torch(rq, evq), rh.DevicePoly.from_torch(rp, evp)
mk = lambda: rh.DevicePoly.from_torch(rq, rb(B, QI60[:24]))
ctA, ctB = rh.Ciphertext([mk(), mk()], is_ntt=True), rh.Ciphertext([mk(), mk()], is_ntt=True)
ctO, ctR = rh.Ciphertext([mk(), mk()], is_ntt=True), rh.Ciphertext([mk(), mk()], is_ntt=True)
if what == "mulrelin":
    cev = rh.ckks.Evaluator(rq, rp, rlk=gct)
    def f():
        cev.MulRelin(ctA, ctB, ctO, relin=True); cev.Rescale(ctO, ctR)
elif what == "rotation":
    kev = rh.rlwe.Evaluator(rq, rp, galois_keys={5: gct})
    dec = kev.DecomposeNTT(23, 5, ctA.Value[1], True)
    f = lambda: kev.AutomorphismHoisted(23, ctA, dec, 5, ctO)
elif what == "decompose":
    kev = rh.rlwe.Evaluator(rq, rp, galois_keys={5: gct})
    dec = kev.DecomposeNTT(23, 5, ctA.Value[1], True)
    f = lambda: kev.DecomposeNTT(23, 5, ctA.Value[1], True, dec)
else:
    be = rh.BasisExtender(rq, rp)
    pP = rh.DevicePoly.from_torch(rp, rb(B, PI60[:6]))
    f = lambda: be.ModDownQPtoQNTT(23, 5, ctA.Value[0], pP, ctO.Value[0])
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(3): f()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print(what, "batch", B, "ms", round(ms, 3), "per s", round(B / ms * 1e3))
