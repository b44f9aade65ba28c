#!/usr/bin/env python3
"""element-wise op timing (A/B aid): bench_vec.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60
N, L, B = 1 << 16, 16, 512
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
if len(sys.argv) > 1: ring.set_tuning("nt_streams", int(sys.argv[1]))     # bench_vec.py 0: default cache policy (A/B)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
mk = lambda: rh.DevicePoly.from_torch(ring, torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs)
pa, pb, pc = mk(), mk(), mk()
out = []
for op in ("ADD", "MUL_MONT", "MFORM", "MUL_MONT_THEN_ADD"):
    f = lambda: ring.vec_op(op, pa, pb, pc)
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(10): f()
    e1.record(stream); torch.cuda.synchronize()
    out.append("%s %.4f" % (op, e0.elapsed_time(e1) / 10))
f = lambda: ring.AutomorphismNTT(pa, 5, pc)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(10): f()
e1.record(stream); torch.cuda.synchronize()
out.append("AUTO %.4f" % (e0.elapsed_time(e1) / 10))
print(" ".join(out))
