#!/usr/bin/env python3
"""bench.py -- forward NTT throughput at the BASELINE metric size (N = 2^16, 16 RNS limbs) on N MI355X GPUs.

A step = Ring.NTT (forward, canonical output: ring/ntt.go:127-131) over one device-resident batch of `--batch` polys
of 16 limbs (Qi60[0:16], ring/test_params.go:15-22), in place.  value = polys transformed per second over all ranks
(weak scaling: every rank owns its own batch; limbs and polys are independent so there is no data-path collective).

One JSON line on stdout (rank 0).  `roofline` is measured live with HIP events on the launch stream over the timed
region; `cpu_baseline` times the oracle's C restatement of nttUnrolled16Lazy+reducevec on the host cores (rank 0, N=1).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QI60 = [0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001, 0x1fffffffff500001,
        0x1fffffffff380001, 0x1fffffffff000001, 0x1ffffffffef00001, 0x1ffffffffee80001,
        0x1ffffffffeb40001, 0x1ffffffffe780001, 0x1ffffffffe600001, 0x1ffffffffe4c0001,
        0x1ffffffffdf40001, 0x1ffffffffdac0001, 0x1ffffffffda40001, 0x1ffffffffc680001]
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--batch", type=int, default=1024, help="polys per GPU per step (1024 -> 8 GiB, >> 256 MiB Infinity Cache)")
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--limbs", type=int, default=16)
    ap.add_argument("--chunk", type=int, default=-1, help="polys per (column,tile) kernel pair; -1 = engine default")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--persistent", type=int, default=-1)
    ap.add_argument("--cluster", type=int, default=-1)
    ap.add_argument("--cluster-wgs", type=int, default=-1)
    ap.add_argument("--unsafe", type=int, default=0)
    ap.add_argument("--order-mix", type=int, default=-1)
    ap.add_argument("--lds-pad", type=int, default=0, help="experiment: extra dynamic LDS bytes per workgroup of the fused launch (occupancy sensitivity)")
    ap.add_argument("--asm-cols", type=int, default=-1, help="hand-scheduled column stages at N=2^16 (default on)")
    ap.add_argument("--prefetch", type=int, default=-1, help="fused launch with tile loads ahead of the column stages (default off: measured no gain)")
    ap.add_argument("--cols2", type=int, default=-1)
    ap.add_argument("--group", type=int, default=-1, help="polys per group of the persistent pipeline")
    ap.add_argument("--asm", type=int, default=-1, help="1/0: hand-scheduled vs C++ forward tile kernel; -1 = engine default")
    args = ap.parse_args()

    import numpy as np
    import torch
    import matrix_fhe_lattigo_amd as rh

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    dist = None
    if args.single_device:
        local_rank = 0
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=torch.device("cuda", local_rank))
        else:
            dist.init_process_group(backend=args.dist_backend)
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)

    N, L, B = 1 << args.logn, args.limbs, args.batch
    mods = QI60[:L]
    ring = rh.Ring(N, mods, device=local_rank)
    stream = torch.cuda.current_stream()
    ring.set_stream(stream.cuda_stream)
    if args.chunk >= 0:
        ring.set_tuning("chunk_polys", args.chunk)
    if args.asm >= 0:
        ring.set_tuning("asm_tile", args.asm)
    if args.cluster >= 0:
        ring.set_tuning("cluster", args.cluster)
    if args.cluster_wgs >= 1:
        ring.set_tuning("cluster_wgs_per_cu", args.cluster_wgs)
    if args.persistent >= 0:
        ring.set_tuning("persistent", args.persistent)
    if args.cols2 >= 0:
        ring.set_tuning("cols2", args.cols2)
    if args.lds_pad > 0:
        ring.set_tuning("dbg_lds_pad", args.lds_pad)
    if args.asm_cols >= 0:
        ring.set_tuning("asm_cols", args.asm_cols)
    if args.prefetch >= 0:
        ring.set_tuning("prefetch", args.prefetch)
    if args.order_mix >= 0:
        ring.set_tuning("order_mix", args.order_mix)
    if args.unsafe:
        ring.set_tuning("persist_unsafe_timing", 1)
    if args.group >= 1:
        ring.set_tuning("group_polys", args.group)

    # synthetic input: i.i.d. residues in [0, q_i), seeded, generated on the device
    g = torch.Generator(device=dev); g.manual_seed(0x5eed + rank)
    data = torch.empty((B, L, N), dtype=torch.int64, device=dev)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    for b0 in range(0, B, 64):
        blk = torch.randint(0, 1 << 62, (min(64, B - b0), L, N), dtype=torch.int64, device=dev, generator=g)
        data[b0:b0 + blk.shape[0]] = blk % qs
    poly = rh.DevicePoly.from_torch(ring, data)

    def step():
        ring.NTT(poly, poly)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    ring.sync()                                   # raises if the pipeline reported a hand-off time-out
    wall = t1 - t0
    dev_ms = e0.elapsed_time(e1)
    if dist is not None:
        t = torch.tensor([wall, dev_ms], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        wall, dev_ms = float(t[0]), float(t[1])

    # per-kernel device time (HIP events on the same stream), outside the timed region
    kern = {}
    if rank == 0 and args.logn > 12:
        for name, phase in (("ntt_fwd_cols", 1), ("ntt_fwd_tile", 2)):
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            ring.ntt_phase(poly, poly, phase=phase)
            torch.cuda.synchronize()
            reps = max(3, min(args.steps, 10))
            a.record(stream)
            for _ in range(reps):
                ring.ntt_phase(poly, poly, phase=phase)
            b.record(stream)
            torch.cuda.synchronize()
            kern[name] = a.elapsed_time(b) / reps

    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = wall * 1e3 / args.steps
    value = world * B * args.steps / wall
    alg_bytes = 16.0 * N * L * B                        # 16*N bytes per limb transform (8 in + 8 out)
    launch_ms = dev_ms / args.steps                     # device time of one whole forward transform of the batch
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic = None
    tpath = os.path.join(ROOT, "profiles", "latest_traffic.json")
    if os.path.exists(tpath):
        try:                                            # PMC-measured bytes (profiles/), valid for the profiled shape only
            tj = json.load(open(tpath))
            if tj.get("config", {}).get("logn") == args.logn and tj.get("config", {}).get("limbs") == L:
                traffic = tj["hbm_bytes_per_poly"] * B
        except Exception:
            traffic = None
    # what is actually launched inside the timed region (engine.hip: rh_std_ntt_launch)
    chunk = args.chunk if args.chunk >= 0 else (max(1, 2048 // L) if B > max(1, 2048 // L) else 0)   # engine.hip auto rule
    special = args.cluster == 1 or args.persistent == 1
    if special:
        launches, kname = 1, "ntt_fwd_cluster / ntt_fwd_persistent (experimental single launch)"
    elif args.logn > 12 and chunk > 0 and B > chunk:
        launches = -(-B // chunk) + 1
        kname = ("ntt_fwd_fused_asm<%d>: column stages of one %d-poly span + tile stages of the previous span; %d launches per step "
                 "(first and last carry one half), each moves the algorithmic bytes of one span's whole transform" % (args.logn - 12, chunk, launches))
    else:
        launches, kname = (2 if args.logn > 12 else 1), "ntt_fwd_cols + ntt_fwd_tile_asm (two launches = one forward transform)"
    out = {
        "metric": "forward-NTT/s at N=2^16, 16 RNS limbs; achieved HBM GB/s vs peak",
        "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic",
        "config": {"workload": "Ring.NTT forward, N=2^%d, %d limbs (Qi60[0:%d]), batch %d polys/GPU, in place, device-resident" % (args.logn, L, L, B),
                   "parallelism": "batch-shard x%d, no collective" % world, "limb_ntt_per_s": value * L},
        "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
                     "traffic": (traffic / launches) if traffic is not None else None,      # PMC bytes per launch, like algorithmic_bytes_per_launch
                     "traffic_per_step": traffic, "kernel": kname, "launches_per_step": launches,
                     "avg_launch_ms": launch_ms / launches, "algorithmic_bytes_per_launch": alg_bytes / launches,
                     "algorithmic_bytes_per_step": alg_bytes, "device_ms_per_step": launch_ms,
                     "standalone_kernel_ms": kern},
    }
    if world == 1 and not args.no_cpu:
        import oracle
        ncpu = os.cpu_count() or 1
        t_probe = oracle.time_ntt_forward(N, mods, 1, 1)            # one 16-limb poly, one thread
        reps = max(1, int(args.cpu_seconds / max(t_probe, 1e-6)))
        t_cpu = oracle.time_ntt_forward(N, mods, reps, 1)
        out["cpu_baseline"] = {"value": reps / t_cpu, "unit": "NTT/s", "cores": 1, "kind": "port",
                               "sample": "%d forward NTTs of one 16-limb N=2^%d poly, single thread (the Go loop is single-threaded), "
                                         "C restatement of nttUnrolled16Lazy+reducevec (oracle/ring_oracle.c), host has %d cores" % (reps, args.logn, ncpu)}
        # the same work spread over this process's share of the host cores (limbs of a poly are independent): what a
        # goroutine-per-limb caller of the reference could reach; a short extra sample, reported beside the like-for-like one
        nthr = min(len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else ncpu, L)
        if nthr > 1:
            reps_mt = max(nthr, int(0.4 * args.cpu_seconds * nthr / max(t_probe, 1e-6)))
            t_mt = oracle.time_ntt_forward(N, mods, reps_mt, nthr)
            out["cpu_baseline"]["all_cores"] = {"value": reps_mt / t_mt, "unit": "NTT/s", "cores": nthr,
                                                "sample": "%d forward NTTs, limbs spread over %d threads" % (reps_mt, nthr)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
