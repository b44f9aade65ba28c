#!/usr/bin/env python3
"""Latency of the host-pointer seam (PCIe inclusive; never `value` of bench.py): what a Go caller of ring.Ring.NTT sees when it
calls the engine unchanged.  For each shape: the reference's loop over limbs through rh_ntt_forward (one synchronous round trip
per limb), the whole-Poly entry rh_ntt_poly_forward, each with pageable and page-locked limbs, beside the CPU oracle's time for
the same call (C restatement of nttUnrolled16Lazy + reducevec, one thread).  JSON on stdout -> profiles/r03_host_path.json."""
import ctypes as C
import json
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import matrix_fhe_lattigo_amd as rh            # noqa: E402
import oracle                                  # noqa: E402
from bench import QI60, cpu_model             # noqa: E402


def timeit(fn, reps, warm=3):
    for _ in range(warm):
        fn()
    best, tot = 1e9, 0.0
    for _ in range(reps):
        t0 = time.perf_counter()
        fn()
        dt = time.perf_counter() - t0
        best = min(best, dt); tot += dt
    return {"mean_us": tot / reps * 1e6, "min_us": best * 1e6, "reps": reps}


def shape(logn, L, reps):
    N, mods = 1 << logn, QI60[:L]
    ring = rh.Ring(N, mods)
    lib = rh.lib()
    rng = np.random.default_rng(logn)
    a = np.stack([(rng.integers(0, 1 << 62, size=N, dtype=np.uint64) % np.uint64(q)) for q in mods])
    pageable_in = [a[i].copy() for i in range(L)]
    pageable_out = [np.zeros(N, dtype=np.uint64) for _ in range(L)]
    pin_in, pin_out = rh.PinnedBuffer((L, N)), rh.PinnedBuffer((L, N))
    pin_in.array[:] = a
    U64P = rh.ringhip.U64P
    out = {"N": N, "limbs": L, "bytes_each_way": N * L * 8}

    def per_limb(ins, outs):
        ptrs = [(x.ctypes.data_as(U64P), y.ctypes.data_as(U64P)) for x, y in zip(ins, outs)]

        def run():
            for i, (x, y) in enumerate(ptrs):
                rc = lib.rh_ntt_forward(ring._h, i, x, y)
                assert rc == 0
        return run

    def whole(ins, outs):
        pi = (C.c_void_p * L)(*[x.ctypes.data for x in ins]); po = (C.c_void_p * L)(*[y.ctypes.data for y in outs])

        def run():
            rc = lib.rh_ntt_poly_forward(ring._h, L - 1, pi, po, 0)
            assert rc == 0
        return run
    pin_rows_in, pin_rows_out = [pin_in.array[i] for i in range(L)], [pin_out.array[i] for i in range(L)]
    out["per_limb_rh_ntt_forward_pageable"] = timeit(per_limb(pageable_in, pageable_out), reps)
    out["per_limb_rh_ntt_forward_pinned"] = timeit(per_limb(pin_rows_in, pin_rows_out), reps)
    out["whole_poly_rh_ntt_poly_forward_pageable"] = timeit(whole(pageable_in, pageable_out), reps)
    out["whole_poly_rh_ntt_poly_forward_pinned"] = timeit(whole(pin_rows_in, pin_rows_out), reps)
    # results identical on all four routes and equal to the oracle
    srs = [oracle.SubRingConsts(N, q) for q in mods]
    want = np.stack([oracle.ntt(a[i], srs[i]) for i in range(L)])
    out["verified"] = bool(np.array_equal(np.stack(pageable_out), want) and np.array_equal(pin_out.array, want))
    # device-resident transform of the same poly (no PCIe), for scale
    p = rh.DevicePoly.from_numpy(ring, a[None])

    def dev():
        ring.NTT(p, p); ring.sync()
    out["device_resident_one_poly"] = timeit(dev, reps)
    # CPU: the oracle's restatement of the reference loop, one thread
    t = oracle.time_ntt_forward(N, mods, max(3, min(reps, 200)), 1)
    out["cpu_oracle_one_thread_us"] = t / max(3, min(reps, 200)) * 1e6
    for k in ("per_limb_rh_ntt_forward_pageable", "per_limb_rh_ntt_forward_pinned", "whole_poly_rh_ntt_poly_forward_pageable", "whole_poly_rh_ntt_poly_forward_pinned"):
        out[k]["GBps_each_way"] = out["bytes_each_way"] / (out[k]["mean_us"] * 1e-6) / 1e9
    pin_in.free(); pin_out.free(); ring.close()
    return out


def main():
    res = {"what": "host-pointer seam latency per Ring.NTT call (upload + transform + download + synchronise), microseconds",
           "cpu": cpu_model(),
           "shapes": [shape(12, 1, 300), shape(12, 16, 200), shape(15, 16, 100), shape(16, 16, 100), shape(16, 24, 60)]}
    print(json.dumps(res, indent=1))


if __name__ == "__main__":
    main()
