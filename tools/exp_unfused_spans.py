#!/usr/bin/env python3
"""EXPERIMENT: does an Infinity-Cache-resident hand-over between the column and the tile stages pay under the power cap?
The metric batch transformed span by span with the UNFUSED launch pair (column stages of the span, then its tile stages: the tile
stages read what the column stages wrote a few tens of microseconds earlier), against the fused pipeline over 1 GiB spans."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matrix_fhe_lattigo_amd as rh
from bench import QI60

N, L, B = 1 << 16, 16, 1024
dev = torch.device("cuda", 0)
ring = rh.Ring(N, QI60[:L])
stream = torch.cuda.current_stream()
ring.set_stream(stream.cuda_stream)
data = torch.randint(0, 1 << 60, (B, L, N), dtype=torch.int64, device=dev)
full = rh.DevicePoly.from_torch(ring, data)


import re, subprocess, threading


def smi():
    t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    c = re.search(r"sclk clock level: \S+ \((\d+)Mhz\)", t); w = re.search(r"Power \(W\): ([\d.]+)", t)
    return (int(c.group(1)) if c else None, float(w.group(1)) if w else None)


def held(fn, seconds=2.5):
    """sclk / package power while fn is re-launched for `seconds`"""
    out = []
    th = threading.Thread(target=lambda: (time.sleep(1.0), out.append(smi()), out.append(smi())))
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds or th.is_alive():
        fn(); torch.cuda.synchronize()
    th.join()
    return out


def timed(fn, reps=12):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(reps):
        fn()
    e1.record(stream); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps, (time.perf_counter() - t0) * 1e3 / reps


print("fused pipeline (default): %.3f ms device, %.3f ms wall" % timed(lambda: ring.NTT(full, full)), held(lambda: ring.NTT(full, full)))
for span in (2, 4, 8, 16, 32, 64, 128):
    views = [rh.DevicePoly.from_torch(ring, data[b0:b0 + span]) for b0 in range(0, B, span)]
    ring.set_tuning("chunk_polys", 0)                              # no pipelining inside a call: one column launch + one tile launch

    def run():
        for v in views:
            ring.NTT(v, v)
    d, w = timed(run, reps=6)
    print("unfused, spans of %3d polys (%4d MiB): %.3f ms device, %.3f ms wall" % (span, span * 8, d, w), held(run) if span >= 8 else "", flush=True)
ring.set_tuning("chunk_polys", -1)
