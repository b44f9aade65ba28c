#!/usr/bin/env python3
"""conjugate-invariant ring NTT / INTT timing (rocprofv3 aid): bench_ci.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60
B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
N, L = 1 << 16, 16
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
ring = rh.Ring(N, QI60[:L], kind=rh.ConjugateInvariant); ring.set_stream(stream.cuda_stream)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
x = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
p = rh.DevicePoly.from_torch(ring, x)
for chunk in ([int(v) for v in sys.argv[2:]] or [-1]):          # chunk_polys values to compare on the same box: -1 auto (pipelined), 0 two launches
  ring.set_tuning("chunk_polys", chunk)
  print("chunk_polys", chunk)
  for name, f in (("NTT", lambda: ring.NTT(p, p)), ("INTT", lambda: ring.INTT(p, p))):
      f(); torch.cuda.synchronize()
      e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
      e0.record(stream)
      for _ in range(5): f()
      e1.record(stream); torch.cuda.synchronize()
      print("CI", name, "B", B, "ms", round(e0.elapsed_time(e1) / 5, 4))
