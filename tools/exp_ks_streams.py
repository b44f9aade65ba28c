#!/usr/bin/env python3
"""Experiment: one gadget product of B polys on one stream vs two of B/2 on two streams (two ring / extender instances)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import torch
import matrix_fhe_lattigo_amd as rh
from conftest import QI60, PI60
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda", 0)
N = 1 << 16
def rb(n, mods):
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, len(mods), 1)
    return torch.randint(0, 1 << 62, (n, len(mods), N), dtype=torch.int64, device=dev) % qs
evq, evp = rb(8, QI60[:24]), rb(8, PI60[:6])
def inst(stream, n):
    rq, rp = rh.Ring(N, QI60[:24]), rh.Ring(N, PI60[:6])
    for r in (rq, rp): r.set_stream(stream.cuda_stream)
    be = rh.BasisExtender(rq, rp)
    xq = rb(n, QI60[:24]); c0, c1 = torch.zeros_like(xq), torch.zeros_like(xq)
    pq, p0, p1 = (rh.DevicePoly.from_torch(rq, t) for t in (xq, c0, c1))
    return (lambda: be.GadgetProduct(23, 5, pq, evq.data_ptr(), evp.data_ptr(), 4, p0, p1)), (rq, rp, be, xq, c0, c1)
s0, s1, s2 = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
f_all, keep0 = inst(s0, B)
f_a, keep1 = inst(s1, B // 2)
f_b, keep2 = inst(s2, B // 2)
for f in (f_all, f_a, f_b): f()
torch.cuda.synchronize()
import time
def wall(fn, reps=5):
    torch.cuda.synchronize(); t = time.perf_counter()
    for _ in range(reps): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t) / reps * 1e3
def two():
    f_a(); f_b()
for rep in range(3):
    print("one stream, %d polys: %.3f ms | two streams, %d each: %.3f ms" % (B, wall(f_all), B // 2, wall(two)))
# offset start: stream 2 first runs a filler (vec ops on its own input) of about a third of a product
rq2 = keep2[0]; x2 = rh.DevicePoly.from_torch(rq2, keep2[3])
def two_offset():
    f_a()
    for _ in range(12): rq2.Add(x2, x2, rh.DevicePoly.from_torch(rq2, keep2[4]))
    f_b()
filler = wall(lambda: [rq2.Add(x2, x2, rh.DevicePoly.from_torch(rq2, keep2[4])) for _ in range(12)])
for rep in range(2):
    print("two streams with stream 2 delayed by a %.3f ms filler: %.3f ms (filler included)" % (filler, wall(two_offset)))
s4 = [torch.cuda.Stream() for _ in range(4)]
quads = [inst(st, B // 4) for st in s4]
for q in quads: q[0]()
torch.cuda.synchronize()
for rep in range(2):
    print("four streams, %d each: %.3f ms" % (B // 4, wall(lambda: [q[0]() for q in quads])))
