#!/usr/bin/env python3
"""What the 1400 W package cap leaves of the shader clock under each kernel of the headline transform (MI355X).

For each workload -- the whole forward transform (fused launches), its column stages alone, its tile stages alone, an element-wise
ADD (pure streaming) -- the kernel is re-launched for ~4 s while `rocm-smi --showclocks --showpower` is sampled from this process
(a child process that makes no HIP call), and the per-call device time of the last second is measured with HIP events.
Output: one line per workload with sclk, package power and ms per call.  Run on the GPU box: python tools/exp_power.py"""
import os
import re
import subprocess
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                                    # noqa: E402
import matrix_fhe_lattigo_amd as rh                             # noqa: E402
from bench import QI60                                          # noqa: E402


def smi():
    t = subprocess.run(["rocm-smi", "--showclocks", "--showpower"], capture_output=True, text=True).stdout
    sclk = re.search(r"sclk clock level: \d+: \((\d+)Mhz\)", t)
    pw = re.search(r"Power \(W\): ([\d.]+)", t)
    return (int(sclk.group(1)) if sclk else None, float(pw.group(1)) if pw else None)


def main():
    N, L, B = 1 << 16, 16, 1024
    dev = torch.device("cuda", 0)
    ring = rh.Ring(N, QI60[:L])
    stream = torch.cuda.current_stream()
    ring.set_stream(stream.cuda_stream)
    g = torch.Generator(device=dev); g.manual_seed(1)
    data = torch.empty((B, L, N), dtype=torch.int64, device=dev)
    qs = torch.tensor(QI60[:L], dtype=torch.int64, device=dev).view(1, L, 1)
    for b0 in range(0, B, 64):
        data[b0:b0 + 64] = torch.randint(0, 1 << 62, (64, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    poly = rh.DevicePoly.from_torch(ring, data)
    other = rh.DevicePoly.from_torch(ring, data.clone())
    work = {
        "idle": None,
        "Ring.NTT (fused column + tile launches)": lambda: ring.NTT(poly, poly),
        "column stages alone (ntt_fwd_cols_asm<4>)": lambda: ring.ntt_phase(poly, poly, phase=1),
        "tile stages alone (ntt_fwd_tile_asm)": lambda: ring.ntt_phase(poly, poly, phase=2),
        "vec ADD (3 streams)": lambda: ring.Add(poly, other, other),
        "Ring.INTT": lambda: ring.INTT(poly, poly),
    }
    # config 5 gadget product (Q = 24, P = 6, batch 64) and the rescale of the metric ring (512 polys, 16 -> 15 limbs)
    from bench import PI60
    rq, rp = rh.Ring(N, QI60[:24]), rh.Ring(N, PI60[:6])
    for r in (rq, rp):
        r.set_stream(stream.cuda_stream)
    be = rh.BasisExtender(rq, rp)

    def rb(n, mods):
        m = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, len(mods), 1)
        return torch.randint(0, 1 << 62, (n, len(mods), N), dtype=torch.int64, device=dev, generator=g) % m
    xq, evq, evp = rb(64, QI60[:24]), rb(8, QI60[:24]), rb(8, PI60[:6])
    c0, c1 = torch.zeros_like(xq), torch.zeros_like(xq)
    pq, p0, p1 = (rh.DevicePoly.from_torch(rq, t) for t in (xq, c0, c1))
    work["GadgetProduct, batch 64 (config 5)"] = lambda: be.GadgetProduct(23, 5, pq, evq.data_ptr(), evp.data_ptr(), 4, p0, p1)
    xp = rb(64, PI60[:6]); oq, op_ = torch.zeros_like(xq), torch.zeros_like(xp)
    pp, poq, pop = rh.DevicePoly.from_torch(rp, xp), rh.DevicePoly.from_torch(rq, oq), rh.DevicePoly.from_torch(rp, op_)
    work["DecomposeAndSplit, one digit, batch 64 (bext_kernel<6>)"] = lambda: be.DecomposeAndSplit(23, 5, 6, 1, pq, poq, pop)
    work["ModUpPtoQ 6 -> 24 limbs, batch 64"] = lambda: be.ModUpPtoQ(5, 23, pp, poq)
    half = rh.DevicePoly.from_torch(ring, data[:512])
    resc = rh.DevicePoly.from_torch(ring.AtLevel(L - 2), torch.empty((512, L - 1, N), dtype=torch.int64, device=dev))
    work["DivRoundByLastModulusNTT, 512 polys"] = lambda: ring.DivRoundByLastModulusNTT(half, resc)
    print("%-44s | sclk MHz (samples) | package W (samples) | ms per call" % "workload")
    for name, fn in work.items():
        if fn is None:
            time.sleep(2.0)
            s = [smi() for _ in range(2)]
            print("%-44s | %s | %s | -" % (name, [x[0] for x in s], [x[1] for x in s]), flush=True)
            continue
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream); fn(); e1.record(stream); torch.cuda.synchronize()
        ms1 = e0.elapsed_time(e1)
        n = max(4, int(4000.0 / ms1))
        samples = []
        stop = threading.Event()

        def sampler():
            time.sleep(1.5)
            while not stop.is_set() and len(samples) < 4:
                samples.append(smi())
                time.sleep(0.3)
        th = threading.Thread(target=sampler); th.start()
        tail = max(1, n // 4)
        for i in range(n):
            if i == n - tail:
                e0.record(stream)
            fn()
            if i % 16 == 15:
                torch.cuda.synchronize()                       # keep the launch queue short so the sampler's timing is honest
        e1.record(stream)
        torch.cuda.synchronize()
        stop.set(); th.join()
        print("%-44s | %s | %s | %.3f" % (name, [x[0] for x in samples], [x[1] for x in samples], e0.elapsed_time(e1) / tail), flush=True)
    ring.close()


if __name__ == "__main__":
    main()
