#!/usr/bin/env python3
"""DivRoundByLastModulusNTT timing (for rocprofv3 --kernel-trace --stats): bench_rescale.py [L] [B]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60
L = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 512
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
N = 1 << 16
ring = rh.Ring(N, QI60[:L]); ring.set_stream(stream.cuda_stream)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
a = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
pa = rh.DevicePoly.from_torch(ring, a)
po = rh.DevicePoly.from_torch(ring, torch.empty((B, L - 1, N), dtype=torch.int64, device=dev))
f = lambda: ring.DivRoundByLastModulusNTT(pa, po)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(5): f()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 5
print("L", L, "batch", B, "ms", round(ms, 3), "poly/s", round(B / ms * 1e3))
