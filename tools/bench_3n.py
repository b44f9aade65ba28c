#!/usr/bin/env python3
"""3N transform timing: bench_3n.py <log2(N/3)> <limbs> <batch> [block_order 0/1]   (profiling aid: rocprofv3 --kernel-trace --stats -- python3 tools/bench_3n.py 13 1 1024)"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch
import matrix_fhe_lattigo_amd as rh
from primes3n import moduli_3n
a, L, B = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
N = 3 << a
mods = moduli_3n(N, L)
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
ring = rh.Ring(N, mods, kind=rh.Matrix3N); ring.set_stream(stream.cuda_stream)
if len(sys.argv) > 4:
    ring.set_tuning("ntt3n_block_order", int(sys.argv[4]))     # 1: device NTT domain in block order (no permutation pass)
if len(sys.argv) > 5:
    ring.set_tuning("perm_fwd_shape", int(sys.argv[5]))        # 10*A + B
if len(sys.argv) > 6:
    ring.set_tuning("perm_inv_shape", int(sys.argv[6]))
qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
x = torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev) % qs
p = rh.DevicePoly.from_torch(ring, x)
for name, f in (("NTT", lambda: ring.NTT(p, p)), ("INTT", lambda: ring.INTT(p, p))):
    f(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(5): f()
    e1.record(stream); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 5
    print("3N", name, "N=%d L=%d B=%d" % (N, L, B), "ms", round(ms, 4), "GB/s(alg)", round(16.0 * N * L * B / ms / 1e6))
