#!/usr/bin/env python3
"""EXPERIMENT: the one-pass forward transform through L2-resident exchange buffers (csrc/ntt_team.hip.hpp, tuning keys team_slots /
team_wgs) against the two-pass pipeline: bit-identical output first (small batch, then the metric batch), then time per step.
usage: python tools/exp_team.py [check|time] ..."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matrix_fhe_lattigo_amd as rh
from bench import QI60

N, L = 1 << 16, 16
dev = torch.device("cuda", 0)
ring = rh.Ring(N, QI60[:L])
stream = torch.cuda.current_stream()
ring.set_stream(stream.cuda_stream)
qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)


def batch(B, seed):
    g = torch.Generator(device=dev); g.manual_seed(seed)
    d = torch.empty((B, L, N), dtype=torch.int64, device=dev)
    for b0 in range(0, B, 64):
        n = min(64, B - b0)
        d[b0:b0 + n] = torch.randint(0, 1 << 62, (n, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    return d


def check(B, S, wgs):
    x = batch(B, 7)
    ref = x.clone(); got = x.clone()
    ring.set_tuning("team_slots", 0)
    pr = rh.DevicePoly.from_torch(ring, ref); ring.NTT(pr, pr)
    ring.set_tuning("team_slots", S); ring.set_tuning("team_wgs", wgs)
    pg = rh.DevicePoly.from_torch(ring, got); ring.NTT(pg, pg)
    err = ring.team_error()
    same = bool(torch.equal(ref, got))
    print("check B=%d S=%d wgs=%d: identical=%s timeout_flag=%s" % (B, S, wgs, same, err), flush=True)
    ring.set_tuning("team_slots", 0)
    return same and not err


def timed(B, S, wgs, reps=8):
    x = batch(B, 9)
    p = rh.DevicePoly.from_torch(ring, x)
    ring.set_tuning("team_slots", S); ring.set_tuning("team_wgs", wgs)
    ring.NTT(p, p); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(reps):
        ring.NTT(p, p)
    e1.record(stream); torch.cuda.synchronize()
    err = ring.team_error() if S else False
    ring.set_tuning("team_slots", 0)
    print("time B=%d S=%d wgs=%d: %.3f ms per step%s" % (B, S, wgs, e0.elapsed_time(e1) / reps, "  (TIMEOUT FLAG)" if err else ""), flush=True)


mode = sys.argv[1] if len(sys.argv) > 1 else "check"
if mode == "check":
    ok = check(8, 9, 4) and check(64, 9, 4) and check(64, 5, 4) and check(64, 6, 3)
    print("ALL OK" if ok else "MISMATCH")
    sys.exit(0 if ok else 1)
elif mode == "one":
    timed(1024, int(sys.argv[2]), int(sys.argv[3]), reps=4)
else:
    timed(1024, 0, 4)
    for S, wgs in ((9, 4), (6, 4), (5, 4), (4, 4), (7, 3), (5, 3), (4, 3), (3, 3)):
        timed(1024, S, wgs)
    timed(1024, 0, 4)
