#!/usr/bin/env python3
"""config 3 poly-mul timing vs the span size of the pipelined launches: exp_config3.py"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
N, L, B = 1 << 15, 16, 512
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
mk = lambda: torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
a, b = mk(), mk()
pa, pb = rh.DevicePoly.from_torch(ring, a), rh.DevicePoly.from_torch(ring, b)
def f_sep():
    ring.NTT(pa, pa); ring.NTT(pb, pb); ring.INTTMul(pa, pb, pa)
def f_many():
    ring.NTTMany([(pa, pa), (pb, pb)]); ring.INTTMul(pa, pb, pa)
for span in (1024, 2048, 4096):
    ring.set_tuning("auto_span_rows", span)
    for name, f in (("two NTT calls", f_sep), ("NTTMany", f_many), ("two NTT calls", f_sep), ("NTTMany", f_many)):
        f(); f(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(10): f()
        e1.record(stream); torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("span_rows", span, name, "ms", round(ms, 3), "polymul/s", round(B / ms * 1e3))
