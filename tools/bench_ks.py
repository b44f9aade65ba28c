#!/usr/bin/env python3
"""Config-5 gadget product timing: bench_ks.py <auto_span_rows> [batch] [graph]
`graph`: also replay the call from a HIP graph captured through torch (latency of small batches)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60, PI60
span, B = int(sys.argv[1]), int(sys.argv[2]) if len(sys.argv) > 2 else 64
dev = torch.device("cuda", 0); stream = torch.cuda.current_stream()
N = 1 << 16
rq, rp = rh.Ring(N, QI60[:24]), rh.Ring(N, PI60[:6])
for r in (rq, rp):
    r.set_stream(stream.cuda_stream); r.set_tuning("auto_span_rows", span)
for kv in os.environ.get("RH_KS_TUNE", "").split(","):          # RH_KS_TUNE=ks_small_rows=0[,key=value...]: ring tunings for the whole run
    if "=" in kv:
        rq.set_tuning(kv.split("=")[0], int(kv.split("=")[1])); rp.set_tuning(kv.split("=")[0], int(kv.split("=")[1]))
be = rh.BasisExtender(rq, rp)
def rb(n, mods):
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, len(mods), 1)
    return torch.randint(0, 1 << 62, (n, len(mods), N), dtype=torch.int64, device=dev) % qs
xq = rb(B, QI60[:24]); evq, evp = rb(8, QI60[:24]), rb(8, PI60[:6])
c0, c1 = torch.zeros_like(xq), torch.zeros_like(xq)
pq, p0, p1 = (rh.DevicePoly.from_torch(rq, t) for t in (xq, c0, c1))
f = lambda: be.GadgetProduct(23, 5, pq, evq.data_ptr(), evp.data_ptr(), 4, p0, p1)
f(); torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(stream)
for _ in range(3): f()
e1.record(stream); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 3
print("span_rows", span, "batch", B, "ms", round(ms, 3), "keyswitch/s", round(B / ms * 1e3))
if os.environ.get("RH_KS_AB"):          # same-box A/B of a ring tuning key: RH_KS_AB=key
    key = os.environ["RH_KS_AB"]
    for rep in range(3):
        for val in [int(v) for v in os.environ.get('RH_KS_AB_VALS', '0,1').split(',')]:
            rq.set_tuning(key, val); rp.set_tuning(key, val)
            f(); torch.cuda.synchronize()
            e0.record(stream)
            for _ in range(5): f()
            e1.record(stream); torch.cuda.synchronize()
            print("  %s=%d ms %.3f" % (key, val, e0.elapsed_time(e1) / 5))
if len(sys.argv) > 3:
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        for r in (rq, rp):
            r.set_stream(side.cuda_stream)
        f(); side.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g, stream=side):
            f()
        g.replay(); side.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(side)
        for _ in range(10): g.replay()
        b.record(side); side.synchronize()
        a2, b2 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a2.record(side)
        for _ in range(10): f()
        b2.record(side); side.synchronize()
        print("  graph replay ms", round(a.elapsed_time(b) / 10, 3), "| direct calls ms", round(a2.elapsed_time(b2) / 10, 3))
