#!/usr/bin/env python3
"""bench.py -- the ring hot path on N MI355X GPUs of one node, one process per GPU.

--workload ntt (default, the BASELINE metric): a step = Ring.NTT (forward, canonical output: ring/ntt.go:127-131) over one
  device-resident batch of `--batch` polys of 16 limbs (Qi60[0:16], ring/test_params.go:15-22) at N = 2^16, in place.
  value = polys transformed per second over all ranks (weak scaling: every rank owns its own batch; limbs and polys are
  independent, so there is no data-path collective -- torch.distributed only carries the barrier and the max-over-ranks).
--workload keyswitch (BASELINE config 5): a step = rlwe.Evaluator.GadgetProduct (core/rlwe/evaluator_gadget_product.go:16-30)
  of `--batch` polys at N = 2^16, Q = Qi60[0:24], P = Pi60[0:6].  --shard limb (default, BASELINE's wording): LIMB-sharded over the
  ranks (rh_kshard_gadget_product behind the C ABI; the one path with a real exchange step: all-gather of the source limbs and of the P
  part, RCCL over xGMI, chunked so that an exchange runs under the other chunk's arithmetic).  --shard batch: the batch divided, the 120 MiB
  key replicated, no collective (DESIGN.md 7 prices both).  value = key switches per second of the whole job (strong scaling).
--workload polymul (BASELINE config 3): a step = c = INTT(NTT(a) . NTT(b)) over `--batch` poly pairs of 16 limbs at N = 2^15 (Ring.PolyMul: the sequence
  NTT, NTT, MForm, MulCoeffsMontgomery, INTT of schemes/ckks/evaluator.go:821-834); batch-sharded like the metric, no data-path collective.
--workload ctmul (BASELINE config 4): a step = matrix_ckks.Evaluator.Mul on `--batch` ciphertext pairs (default 128 = one GPU's share of 1024) over the 3N ring
  N = 3*2^16 (`--logn 14`: N = 3*2^14), 24 limbs; batch-sharded, no data-path collective.
--gather (ntt workload): after the timed region, the north star's one collective -- the final gather of `--gather-polys` result polys per
  rank as ONE all_gather_into_tensor on device memory (sharding.gather_polys) -- timed on its own and reported OUTSIDE `value`.

`python bench.py --gpus N` with N > 1 and no WORLD_SIZE in the environment starts the N ranks itself (fresh processes, before
any GPU call) and relays rank 0's JSON line; under torchrun (WORLD_SIZE set) it is one rank.

One JSON line on stdout (rank 0).  `roofline` is measured live with HIP events on the launch stream over the timed region;
`verified` says the timed region's own output was checked against the CPU oracle after timing; `cpu_baseline` times the
oracle's C restatement of nttUnrolled16Lazy + reducevec on the host cores (rank 0, N = 1)."""
import argparse
import json
import os
import socket
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

QI60 = [0x1fffffffffe00001, 0x1fffffffffc80001, 0x1fffffffffb40001, 0x1fffffffff500001,
        0x1fffffffff380001, 0x1fffffffff000001, 0x1ffffffffef00001, 0x1ffffffffee80001,
        0x1ffffffffeb40001, 0x1ffffffffe780001, 0x1ffffffffe600001, 0x1ffffffffe4c0001,
        0x1ffffffffdf40001, 0x1ffffffffdac0001, 0x1ffffffffda40001, 0x1ffffffffc680001,
        0x1ffffffffc000001, 0x1ffffffffb880001, 0x1ffffffffb7c0001, 0x1ffffffffb300001,
        0x1ffffffffb1c0001, 0x1ffffffffadc0001, 0x1ffffffffa400001, 0x1ffffffffa140001]
PI60 = [0x1ffffffff6c80001, 0x1ffffffff6140001, 0x1ffffffff5f40001, 0x1ffffffff5700001,
        0x1ffffffff4bc0001, 0x1ffffffff4380001]
HBM_PEAK_GBS = 8000.0     # MI355X HBM3E spec peak (/opt/skills/guides/MI355X_MICROARCH.md)


def parse():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--workload", default="ntt", choices=["ntt", "keyswitch", "polymul", "ctmul"])
    ap.add_argument("--batch", type=int, default=-1, help="ntt: polys per GPU per step (default 1024 -> 8 GiB, >> 256 MiB Infinity Cache); "
                                                           "keyswitch: polys per step of the whole job (default 64)")
    ap.add_argument("--logn", type=int, default=16)
    ap.add_argument("--limbs", type=int, default=16, help="ntt workload: limbs of the ring (Qi60[0:limbs])")
    ap.add_argument("--chunk", type=int, default=-1, help="polys per (column,tile) kernel pair; -1 = engine default")
    ap.add_argument("--cpu-seconds", type=float, default=10.0, help="target wall time of the CPU baseline sample")
    ap.add_argument("--no-cpu", action="store_true")
    ap.add_argument("--no-power", action="store_true", help="skip the clock / package-power sample under load (rocm-smi)")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed region's output")
    ap.add_argument("--dist-backend", default="nccl", help="nccl (= RCCL, default) or gloo (rehearsal of the N>1 path on one GPU)")
    ap.add_argument("--single-device", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--asm-cols", type=int, default=-1, help="hand-scheduled column stages (default on)")
    ap.add_argument("--asm", type=int, default=-1, help="1/0: hand-scheduled vs C++ tile kernels; -1 = engine default")
    ap.add_argument("--tune", action="append", default=[], help="key=value for rh_ring_set_tuning (repeatable)")
    ap.add_argument("--shard", default="limb", choices=["limb", "batch"], help="keyswitch workload: limb-shard (exchange steps) or batch-shard (key replicated, no collective)")
    ap.add_argument("--chunks", type=int, default=0, help="keyswitch --shard limb: chunks of the batch pipelined on two streams (0 = auto: 4 on more than one rank)")
    ap.add_argument("--gather", action="store_true", help="ntt workload: time the final device-tensor gather of result polys (outside `value`)")
    ap.add_argument("--gather-polys", type=int, default=64, help="polys per rank in the --gather leg (64 -> 512 MiB per rank at the default shape)")
    ap.add_argument("--force-dist", action="store_true", help="pre-flight: with one rank, still create the process group and run every collective the N > 1 "
                                                              "path makes (all-reduce, barrier, all-gather of clocks, AND of flags, the --gather leg) -- RCCL with a world of 1")
    ap.add_argument("--rank-timeout", type=float, default=1500.0, help="--gpus N without a launcher: the parent gives up (and stops its ranks) after this many seconds")
    return ap.parse_args()


def csrc_tree_hash():
    """sha256 (16 hex digits) over the kernel sources libringhip.so is built from: ties a counter file under profiles/ to the code it was
    collected on (the GPU box has no .git, so this is a content hash, not `git rev-parse HEAD:.../csrc`)"""
    import hashlib
    h = hashlib.sha256()
    d = os.path.join(ROOT, "matrix-fhe-lattigo_amd", "csrc")
    for name in sorted(os.listdir(d)):
        if name.endswith((".hip", ".hpp", ".inc")):
            h.update(name.encode())
            h.update(open(os.path.join(d, name), "rb").read())
    return h.hexdigest()[:16]


def spawn_ranks(args):
    """--gpus N > 1 without a launcher: start N fresh rank processes (this parent makes no GPU call), relay rank 0's line"""
    s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
    procs = []
    for r in range(args.gpus):
        env = dict(os.environ, RANK=str(r), LOCAL_RANK=str(r), WORLD_SIZE=str(args.gpus), LOCAL_WORLD_SIZE=str(args.gpus),
                   MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
        # this pool's host driver only supports dmabuf IPC: without it RCCL's intra-node buffer exchange (and any CUDA-IPC tensor sharing)
        # fails with `hipIpcGetMemHandle: invalid argument`.  The image exports it already; kept for a shell that dropped the variable.
        env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
        procs.append(subprocess.Popen([sys.executable, os.path.abspath(__file__)] + sys.argv[1:], env=env,
                                      stdout=subprocess.PIPE if r == 0 else subprocess.DEVNULL))
    # relay rank 0's line; if any rank dies, stop the others (their own PIDs) instead of waiting in a collective for the time-out
    import threading
    chunks = []
    reader = threading.Thread(target=lambda: chunks.append(procs[0].stdout.read()), daemon=True)
    reader.start()
    failed = False
    t_start = time.time()
    while True:
        rcs = [p.poll() for p in procs]
        if any(rc not in (None, 0) for rc in rcs):
            failed = True
            break
        if all(rc == 0 for rc in rcs):
            break
        if time.time() - t_start > args.rank_timeout:       # a hang no rank dies of (e.g. a collective one rank never enters)
            sys.stderr.write("bench.py: ranks still running after %.0f s, stopping them\n" % args.rank_timeout)
            failed = True
            break
        time.sleep(0.2)
    if failed:
        for p in procs:
            if p.poll() is None:
                p.terminate()
        for p in procs:
            try:
                p.wait(timeout=30)
            except subprocess.TimeoutExpired:
                p.kill()
    reader.join(timeout=10)
    sys.stdout.write(b"".join(c for c in chunks if c).decode())
    sys.stdout.flush()
    if failed:
        sys.stderr.write("bench.py: rank exit codes %s\n" % [p.returncode for p in procs])
        sys.exit(1)


def init_dist(args):
    import signal
    import torch
    signal.signal(signal.SIGTERM, _on_sigterm)          # see _leave_if_stopped
    rank = int(os.environ.get("RANK", "0"))
    local_rank = 0 if args.single_device else int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if args.gpus > 1 and world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d" % (args.gpus, world))
    dist, ranks_seen = None, 1
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1 or args.force_dist:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if world == 1 and "MASTER_PORT" not in os.environ:          # --force-dist without a launcher: a rendezvous of one
            s = socket.socket(); s.bind(("127.0.0.1", 0)); os.environ["MASTER_PORT"] = str(s.getsockname()[1]); s.close()
            os.environ.setdefault("RANK", "0"); os.environ.setdefault("WORLD_SIZE", "1")
        if args.dist_backend == "nccl":
            dist.init_process_group(backend="nccl", device_id=dev)
        else:
            dist.init_process_group(backend=args.dist_backend)
        assert dist.get_world_size() == world
        one = torch.ones(1, dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
        dist.all_reduce(one)                                   # a real collective over the backend (RCCL when nccl)
        ranks_seen = int(one.item())
        assert ranks_seen == world, "all_reduce of ones returned %d, world is %d" % (ranks_seen, world)
    return rank, local_rank, world, dist, dev, ranks_seen


_STOP = [False]       # set by SIGTERM (the parent stopping its ranks after one of them died): leave at the next step boundary, with the GPU idle


def _on_sigterm(signum, frame):
    _STOP[0] = True


def _leave_if_stopped():
    """a rank told to stop finishes the step it is in, drains its stream and exits: a process must not die with kernels in flight on a shared GPU
    (tearing down queues with running waves is the one thing on this path that can disturb OTHER processes on the card)"""
    if _STOP[0]:
        import torch
        torch.cuda.synchronize()
        sys.stderr.write("bench.py: rank %s stopping on SIGTERM\n" % os.environ.get("RANK", "0"))
        sys.stderr.flush()
        os._exit(143)


def timed_region(step, args, dist, dev, stream):
    """W untimed steps, then exactly K steps bracketed by barrier + synchronize on both sides; MAX over ranks"""
    import torch
    for _ in range(args.warmup):
        _leave_if_stopped()
        step()
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record(stream)
    for _ in range(args.steps):
        if _STOP[0]:
            _leave_if_stopped()
        step()
    e1.record(stream)
    torch.cuda.synchronize()
    if dist is not None:
        dist.barrier()
    torch.cuda.synchronize()
    wall = time.perf_counter() - t0
    dev_ms = e0.elapsed_time(e1)
    per_rank = {"device_ms_per_step": [dev_ms / args.steps]}
    if dist is not None:
        cdev = dev if args.dist_backend == "nccl" else "cpu"
        mine = torch.tensor([wall, dev_ms], dtype=torch.float64, device=cdev)
        every = [torch.empty_like(mine) for _ in range(dist.get_world_size())]
        dist.all_gather(every, mine)                       # every rank's own clock: an imbalanced node shows here
        per_rank = {"device_ms_per_step": [float(t[1]) / args.steps for t in every]}
        wall, dev_ms = max(float(t[0]) for t in every), max(float(t[1]) for t in every)
    per_rank["min"], per_rank["max"] = min(per_rank["device_ms_per_step"]), max(per_rank["device_ms_per_step"])
    return wall, dev_ms, per_rank


def all_ranks_ok(ok, args, dist, dev):
    """AND of a per-rank flag over the job (every rank verifies its own output)"""
    import torch
    if dist is None:
        return bool(ok)
    flag = torch.tensor([1.0 if ok else 0.0], dtype=torch.float64, device=dev if args.dist_backend == "nccl" else "cpu")
    dist.all_reduce(flag, op=dist.ReduceOp.MIN)
    return bool(flag.item() == 1.0)


def cpu_model():
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                return line.split(":", 1)[1].strip()
    except OSError:
        pass
    return "unknown"


def usable_cpus():
    """threads this process can really run at once: the scheduler affinity capped by the cgroup CPU quota (a GPU box gives each
    GPU a share of the host's cores; oversubscribing it only measures the throttle)"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    for path, parse in (("/sys/fs/cgroup/cpu.max", lambda t: t.split()),
                        ("/sys/fs/cgroup/cpu/cpu.cfs_quota_us", lambda t: (t.strip(), open("/sys/fs/cgroup/cpu/cpu.cfs_period_us").read().strip()))):
        try:
            quota, period = parse(open(path).read())
            if quota != "max" and int(quota) > 0:
                n = min(n, max(1, int(quota) // int(period)))
            break
        except (OSError, ValueError):
            continue
    return n


def cpu_baseline(args, N, mods):
    import oracle
    L = len(mods)
    ncpu = os.cpu_count() or 1
    avail = usable_cpus()
    t_probe = oracle.time_ntt_forward(N, mods, 1, 1)            # one L-limb poly, one thread
    reps = max(1, int(args.cpu_seconds / max(t_probe, 1e-6)))
    t_cpu = oracle.time_ntt_forward(N, mods, reps, 1)
    out = {"value": reps / t_cpu, "unit": "NTT/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
           "sample": "%d forward NTTs of one %d-limb N=2^%d poly, single thread (the Go loop over limbs is single-threaded), C restatement of "
                     "nttUnrolled16Lazy+reducevec (oracle/ring_oracle.c); host has %d cores, %d usable by this process (affinity and cgroup quota)" % (reps, L, args.logn, ncpu, avail)}
    if avail > 1:
        # every core this process may use, (poly, limb) units spread over the threads (a goroutine-per-limb caller of the reference
        # over a batch): a short extra sample, reported beside the like-for-like one.  A GPU box may advertise more cores than its
        # share lets run (the quota is not always visible): probe one pass with all advertised cores and with 16, keep the faster.
        best = None
        for nthr in sorted({avail, min(avail, 16)}):
            npolys = max(1, -(-nthr // L))
            t1 = oracle.time_ntt_forward_polys(N, mods, npolys, 1, nthr)
            rate = npolys / max(t1, 1e-9)
            if best is None or rate > best[0]:
                best = (rate, nthr, npolys, t1)
        _, nthr, npolys, t1 = best
        reps_mt = max(1, min(int(0.4 * args.cpu_seconds / max(t1, 1e-6)), 100000))
        t_mt = oracle.time_ntt_forward_polys(N, mods, npolys, reps_mt, nthr)
        out["all_cores"] = {"value": npolys * reps_mt / t_mt, "unit": "NTT/s", "cores": nthr,
                            "sample": "%d passes over %d polys, %d (poly, limb) units spread over %d threads (%d cores usable by this process, "
                                      "%d on the host)" % (reps_mt, npolys, npolys * L, nthr, avail, ncpu)}
    return out


def power_sample(step, dev_index, seconds=1.8):
    """Shader clock and package power WHILE the step runs: the step is re-launched for `seconds` on this thread, `rocm-smi` is run
    from a sampler thread (a child process that makes no HIP call).  None when rocm-smi is not available."""
    import re
    import shutil
    import threading
    import torch
    exe = shutil.which("rocm-smi")
    if exe is None:
        return None

    def smi(*flags):
        try:
            return subprocess.run([exe] + list(flags), capture_output=True, text=True, timeout=20).stdout
        except (OSError, subprocess.SubprocessError):
            return ""
    samples, stop = [], threading.Event()

    def sampler():
        # every GPU rocm-smi sees is read and the one drawing the most power is kept: rocm-smi numbers physical devices, which need
        # not be this process's HIP ordinal (HIP_VISIBLE_DEVICES), and on a shared node the other GPUs idle during an N = 1 run
        time.sleep(0.6 * seconds / 1.8)
        while not stop.is_set() and len(samples) < 3:
            t = smi("--showclocks", "--showpower")
            clk = {int(g): int(c) for g, c in re.findall(r"GPU\[(\d+)\]\s*: sclk clock level: \S+ \((\d+)Mhz\)", t)}
            pw = {int(g): float(w) for g, w in re.findall(r"GPU\[(\d+)\]\s*: [^\n]*Power \(W\): ([\d.]+)", t)}
            both = [g for g in pw if g in clk]
            if both:
                g = max(both, key=lambda k: pw[k])
                samples.append((clk[g], pw[g], g))
    th = threading.Thread(target=sampler)
    th.start()
    t0 = time.perf_counter()
    while time.perf_counter() - t0 < seconds or (th.is_alive() and time.perf_counter() - t0 < 4 * seconds):
        for _ in range(8):
            step()
        torch.cuda.synchronize()
    stop.set(); th.join()
    if not samples:
        return None
    gpu = samples[-1][2]
    cap = re.search(r"GPU\[%d\]\s*: [^\n]*Max Graphics Package Power \(W\): ([\d.]+)" % gpu, smi("--showmaxpower"))
    return {"sclk_mhz_under_load": sum(c for c, _, _ in samples) / len(samples), "package_w_under_load": sum(w for _, w, _ in samples) / len(samples),
            "package_cap_w": float(cap.group(1)) if cap else None, "sclk_peak_mhz": 2400, "samples": len(samples), "rocm_smi_gpu": gpu,
            "source": "rocm-smi --showclocks --showpower sampled while the timed step is re-launched for %.1f s after the timed region" % seconds}


def load_json(name):
    p = os.path.join(ROOT, "profiles", name)
    try:
        return json.load(open(p))
    except (OSError, ValueError):
        return None


# ------------------------------------------------------------------------------------------------ workload: ntt
def run_ntt(args):
    import numpy as np
    import torch
    import matrix_fhe_lattigo_amd as rh
    rank, local_rank, world, dist, dev, ranks_seen = init_dist(args)
    N, L = 1 << args.logn, args.limbs
    B = args.batch if args.batch > 0 else 1024
    mods = QI60[:L]
    ring = rh.Ring(N, mods, device=local_rank)
    stream = torch.cuda.current_stream()
    ring.set_stream(stream.cuda_stream)
    if args.chunk >= 0:
        ring.set_tuning("chunk_polys", args.chunk)
    if args.asm >= 0:
        ring.set_tuning("asm_tile", args.asm)
    if args.asm_cols >= 0:
        ring.set_tuning("asm_cols", args.asm_cols)
    for kv in args.tune:
        k, v = kv.split("=")
        ring.set_tuning(k, int(v))

    # synthetic input: i.i.d. residues in [0, q_i), seeded, generated on the device
    g = torch.Generator(device=dev); g.manual_seed(0x5eed + rank)
    data = torch.empty((B, L, N), dtype=torch.int64, device=dev)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    for b0 in range(0, B, 64):
        blk = torch.randint(0, 1 << 62, (min(64, B - b0), L, N), dtype=torch.int64, device=dev, generator=g)
        data[b0:b0 + blk.shape[0]] = blk % qs
    poly = rh.DevicePoly.from_torch(ring, data)
    spots = sorted({(0, 0), (min(B - 1, B // 2 + 1), min(L - 1, 7)), (B - 1, L - 1)})
    saved = {s: data[s[0], s[1]].cpu().numpy().view(np.uint64).copy() for s in spots}      # every rank verifies its own batch

    def step():
        ring.NTT(poly, poly)

    wall, dev_ms, per_rank = timed_region(step, args, dist, dev, stream)
    ring.sync()
    # verification of the timed region's own output, on EVERY rank: the buffer has been transformed warmup + steps times in place
    verified = None
    if not args.no_verify:
        import oracle
        k = args.warmup + args.steps
        ok = True
        for (p, l), x in saved.items():
            sr = oracle.SubRingConsts(N, mods[l])
            for _ in range(k):
                x = oracle.ntt(x, sr)
            if not np.array_equal(data[p, l].cpu().numpy().view(np.uint64), x):
                ok = False
        if not ok:
            sys.stderr.write("bench.py: rank %d: OUTPUT MISMATCH vs oracle after %d forward transforms\n" % (rank, k))
        verified = all_ranks_ok(ok, args, dist, dev)

    # the final gather of results (north star: "RCCL over xGMI used only for the final gather"), timed on its own, never part of `value`
    gather = None
    if args.gather:
        from matrix_fhe_lattigo_amd import sharding
        gp = max(1, min(args.gather_polys, B))
        block = data[:gp]
        out = torch.empty((world, gp, L, N), dtype=torch.int64, device=dev)
        sharding.gather_polys(block, dist, out=out, force=args.force_dist)          # warm-up (RCCL builds its rings on first use)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            sharding.gather_polys(block, dist, out=out, force=args.force_dist)
        torch.cuda.synchronize()
        if dist is not None:
            dist.barrier()
        g_ms = (time.perf_counter() - t0) * 1e3 / reps
        # what arrived: this rank's slot is its own block; every other slot carries that rank's fingerprint (exchanged apart, tiny)
        fp = block[0, 0, :8].clone() if args.dist_backend == "nccl" else block[0, 0, :8].cpu()     # RCCL moves device tensors, gloo host tensors
        fps = [fp]
        if dist is not None:
            fps = [torch.empty_like(fp) for _ in range(world)]
            dist.all_gather(fps, fp)
        ok = bool(torch.equal(out[rank], block)) and all(bool(torch.equal(out[r, 0, 0, :8].cpu(), fps[r].cpu())) for r in range(world))
        recv = (world - 1) * gp * L * N * 8
        gather = {"ms": g_ms, "polys_per_rank": gp, "bytes_received_per_gpu": recv, "GBps_received_per_gpu": recv / (g_ms * 1e-3) / 1e9 if world > 1 else None,
                  "verified": all_ranks_ok(ok, args, dist, dev),
                  "what": "one all_gather_into_tensor of every rank's (%d, %d, %d) result block (sharding.gather_polys), %s; outside `value`"
                          % (gp, L, N, "RCCL" if args.dist_backend == "nccl" else "host-staged over " + args.dist_backend)}

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
    power = power_sample(step, local_rank) if (rank == 0 and world == 1 and not args.no_power) else None
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return

    ms_per_step = wall * 1e3 / args.steps
    value = world * B * args.steps / wall
    alg_bytes = 16.0 * N * L * B                        # 16*N bytes per limb transform (8 in + 8 out)
    launch_ms = dev_ms / args.steps                     # device time of one whole forward transform of the batch
    achieved = alg_bytes / (launch_ms * 1e-3) / 1e9
    traffic, stale = None, None
    tree = csrc_tree_hash()
    tj = load_json("latest_traffic.json")
    if tj and tj.get("config", {}).get("logn") == args.logn and tj.get("config", {}).get("limbs") == L:
        if tj.get("csrc_tree") == tree:
            traffic = tj["hbm_bytes_per_poly"] * B      # PMC-measured bytes (profiles/), valid for the profiled shape and THESE sources only
        else:
            stale = "profiles/latest_traffic.json was collected on csrc tree %s, this run is %s: refresh with tools/refresh_profiles.sh" % (tj.get("csrc_tree"), tree)
    # what is actually launched inside the timed region (engine.hip: rh_std_ntt_launch)
    span_rows = 2048
    for kv in args.tune:
        if kv.split("=")[0] == "auto_span_rows":
            span_rows = int(kv.split("=")[1])
    span = max(1, span_rows // L)
    chunk = args.chunk if args.chunk >= 0 else (span if B > span else 0)   # engine.hip auto rule
    if args.logn > 12 and chunk > 0 and B > chunk:
        launches = -(-B // chunk) + 1
        kname = ("ntt_fwd_fused_asm<%d>: column stages of one %d-poly span + tile stages of the previous span; %d launches per step "
                 "(first and last carry one half), each moves the algorithmic bytes of one span's whole transform" % (args.logn - 12, chunk, launches))
    else:
        launches, kname = (2 if args.logn > 12 else 1), "ntt_fwd_cols + ntt_fwd_tile_asm (two launches = one forward transform)"
    # "bound": the roofline the fraction is quoted against (the north star's: HBM).  "limited_by": what measurably binds this kernel --
    # the 1400 W package cap under full-rate 64-bit VALU work + 4.8 TB/s of fabric traffic (roofline.power / .valu below)
    roof = {"bound": "hbm", "limited_by": "power/valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS,
            "traffic": (traffic / launches) if traffic is not None else None,      # PMC bytes per launch, like algorithmic_bytes_per_launch
            "traffic_note": stale, "csrc_tree": tree,
            "traffic_per_step": traffic, "kernel": kname, "launches_per_step": launches,
            "avg_launch_ms": launch_ms / launches, "algorithmic_bytes_per_launch": alg_bytes / launches,
            "algorithmic_bytes_per_step": alg_bytes, "device_ms_per_step": launch_ms, "standalone_kernel_ms": kern}
    vj = load_json("latest_valu.json")                  # SQ_INSTS_VALU per step + measured issue peak (profiles/), profiled shape and sources only
    if vj and vj.get("config", {}).get("logn") == args.logn and vj.get("config", {}).get("limbs") == L and vj.get("csrc_tree") != tree:
        roof["valu"] = None
        roof["valu_note"] = "profiles/latest_valu.json was collected on csrc tree %s, this run is %s" % (vj.get("csrc_tree"), tree)
    elif vj and vj.get("config", {}).get("logn") == args.logn and vj.get("config", {}).get("limbs") == L:
        winstr = vj["valu_wave_instructions_per_poly"] * B
        issue = winstr / (launch_ms * 1e-3) / 1e9
        roof["valu"] = {"note": "the transform is VALU-issue bound before it is HBM-bound (DESIGN.md 3): 64-bit modular butterflies, no MFMA-shaped work",
                        "wave_instructions_per_step": winstr, "achieved_Gwinstr_per_s": issue,
                        "issue_peak_Gwinstr_per_s": vj["issue_peak_Gwinstr_per_s"], "frac_of_issue_peak": issue / vj["issue_peak_Gwinstr_per_s"],
                        "butterfly_ceiling_NTT_per_s": vj["butterfly_ceiling_limb_ntt_per_s"] / L,
                        "frac_of_butterfly_ceiling": (B / (launch_ms * 1e-3)) / (vj["butterfly_ceiling_limb_ntt_per_s"] / L),
                        "source": vj.get("source")}
    if power is not None:
        # the package power cap, not a pipeline, sets the clock this kernel runs at: both ceilings above are
        # quoted at the 2.4 GHz the chip holds for pure VALU streams; at the clock measured under THIS kernel they scale by sclk / 2400
        roof["power"] = power
        if traffic is not None and roof.get("valu") and power.get("package_cap_w"):
            # energy floor of this design at the cap (constants measured on this chip, profiles/r02_mem_power_l2_mall.txt,
            # profiles/r02_power_clock_samples.txt): HBM-path traffic 0.137 nJ per byte, a VALU wave-instruction of this mix ~1.1 nJ,
            # 291 W drawn idle; (traffic + arithmetic energy) / (cap - idle) = the time the cap allows for one step
            e_mem, e_valu, idle_w = traffic * 0.137e-9, roof["valu"]["wave_instructions_per_step"] * 1.1e-9, 291.0
            floor_ms = (e_mem + e_valu) / (power["package_cap_w"] - idle_w) * 1e3
            roof["power"]["energy_model"] = {"traffic_J_per_step": e_mem, "valu_J_per_step": e_valu, "idle_w": idle_w,
                                             "floor_ms_per_step_at_cap": floor_ms, "frac_of_floor": floor_ms / launch_ms}
        if roof.get("valu"):
            k = power["sclk_mhz_under_load"] / power["sclk_peak_mhz"]
            roof["valu"]["issue_peak_at_measured_clock_Gwinstr_per_s"] = roof["valu"]["issue_peak_Gwinstr_per_s"] * k
            roof["valu"]["frac_of_issue_peak_at_measured_clock"] = roof["valu"]["frac_of_issue_peak"] / k
    out = {
        "metric": "forward-NTT/s at N=2^%d, %d RNS limbs; achieved HBM GB/s vs peak" % (args.logn, L),
        "value": value, "unit": "NTT/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "verified": verified,
        "config": {"workload": "Ring.NTT forward, N=2^%d, %d limbs (Qi60[0:%d]), batch %d polys/GPU, in place, device-resident" % (args.logn, L, L, B),
                   "parallelism": "batch-shard x%d, no data-path collective" % world, "limb_ntt_per_s": value * L,
                   "dist_backend": args.dist_backend if (world > 1 or args.force_dist) else None, "rccl_ranks" if args.dist_backend == "nccl" else "dist_ranks": ranks_seen,
                   "per_rank_device_ms_per_step": per_rank, "verified_on": "every rank (its own batch, %d spot rows each)" % len(spots)},
        "roofline": roof,
    }
    if gather is not None:
        out["final_gather"] = gather
    if world == 1 and not args.no_cpu:
        out["cpu_baseline"] = cpu_baseline(args, N, mods)
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ workload: polymul (BASELINE config 3)
def run_polymul(args):
    """a step = c = INTT(NTT(a) . NTT(b)) on a batch of polys at N = 2^15, 16 limbs (schemes/ckks/evaluator.go:821-834: NTT, NTT, MForm,
    MulCoeffsMontgomery, INTT) through Ring.PolyMul; batch-sharded like the metric: every rank its own batch, no data-path collective"""
    import numpy as np
    import torch
    import matrix_fhe_lattigo_amd as rh
    rank, local_rank, world, dist, dev, ranks_seen = init_dist(args)
    logn = args.logn if any(a.startswith("--logn") for a in sys.argv) else 15     # config 3's ring degree unless --logn says otherwise
    N, L = 1 << logn, args.limbs
    B = args.batch if args.batch > 0 else 512
    mods = QI60[:L]
    ring = rh.Ring(N, mods, device=local_rank)
    stream = torch.cuda.current_stream()
    ring.set_stream(stream.cuda_stream)
    for kv in args.tune:
        k, v = kv.split("=")
        ring.set_tuning(k, int(v))
    g = torch.Generator(device=dev); g.manual_seed(0xc3 + rank)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)
    mk = lambda: torch.randint(0, 1 << 62, (B, L, N), dtype=torch.int64, device=dev, generator=g) % qs
    ta, tb = mk(), mk()
    tc = torch.empty_like(ta)
    pa, pb, pc = (rh.DevicePoly.from_torch(ring, t) for t in (ta, tb, tc))
    ka, kb = ta.clone(), tb.clone()                            # (PolyMul transforms its operands in place: every step starts from these)

    def step():
        ta.copy_(ka); tb.copy_(kb)
        ring.PolyMul(pa, pb, pc)

    # the two restore copies ride in the timed steps, are timed apart right after, and are subtracted: `value` is the product alone
    wall, dev_ms, per_rank = timed_region(step, args, dist, dev, stream)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(stream)
    for _ in range(args.steps):
        ta.copy_(ka); tb.copy_(kb)
    e1.record(stream); torch.cuda.synchronize()
    copy_ms = e0.elapsed_time(e1)
    ring.sync()
    verified = None
    spots = sorted({(0, 0), (min(B - 1, B // 2 + 1), min(L - 1, 7)), (B - 1, L - 1)})
    if not args.no_verify:
        import oracle
        MUL, MFORM = rh.OPS["MUL_MONT"], rh.OPS["MFORM"]
        ok = True
        z = np.zeros(N, dtype=np.uint64)
        for (p_, l) in spots:
            sr = oracle.SubRingConsts(N, mods[l])
            xa, xb = (t[p_, l].cpu().numpy().view(np.uint64) for t in (ka, kb))
            fa, fb = oracle.ntt(xa, sr), oracle.ntt(xb, sr)
            prod = oracle.vec_op(MUL, oracle.vec_op(MFORM, fa, z, z, 0, 0, mods[l]), fb, z, 0, 0, mods[l])
            if not np.array_equal(tc[p_, l].cpu().numpy().view(np.uint64), oracle.intt(prod, sr)):
                ok = False
        if not ok:
            sys.stderr.write("bench.py: rank %d: POLY-MUL MISMATCH vs oracle\n" % rank)
        verified = all_ranks_ok(ok, args, dist, dev)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    dev_step_ms = (dev_ms - copy_ms) / args.steps               # the product alone (device time); the wall clock keeps the restore copies in
    wall_prod = max(wall - copy_ms * 1e-3, 1e-9)
    value = world * B * args.steps / wall_prod
    seq_bytes, fused_bytes, moved_bytes = 88.0 * N * L * B, 24.0 * N * L * B, 72.0 * N * L * B      # SURVEY 8(d); 72: what Ring.PolyMul moves
    achieved = seq_bytes / (dev_step_ms * 1e-3) / 1e9
    out = {
        "metric": "poly-mul/s at N=2^%d, %d RNS limbs (c = INTT(NTT(a) . NTT(b)), BASELINE config 3)" % (logn, L),
        "value": value, "unit": "poly-mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall_prod * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "verified": verified,
        "config": {"workload": "Ring.PolyMul, N=2^%d, %d limbs (Qi60[0:%d]), batch %d polys/GPU, device-resident; operands restored before every step (their restore copies timed apart and subtracted)" % (logn, L, L, B),
                   "parallelism": "batch-shard x%d, no data-path collective" % world,
                   "dist_backend": args.dist_backend if (world > 1 or args.force_dist) else None, "rccl_ranks" if args.dist_backend == "nccl" else "dist_ranks": ranks_seen,
                   "per_rank_device_ms_per_step": per_rank, "verified_on": "every rank (%d spot rows of its own batch vs the oracle's NTT, NTT, MForm, MulCoeffsMontgomery, INTT)" % len(spots)},
        "roofline": {"bound": "hbm", "limited_by": "power/valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_step": seq_bytes, "note": "algorithmic bytes = the reference's kernel sequence, 88*N*L per product (SURVEY 8(d))",
                     "frac_vs_fused_lower_bound_24NL": fused_bytes / (dev_step_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     "bytes_moved_by_this_implementation_per_step": moved_bytes, "device_ms_per_step": dev_step_ms},
    }
    if world == 1 and not args.no_cpu:
        import oracle
        MUL, MFORM = rh.OPS["MUL_MONT"], rh.OPS["MFORM"]
        srs = [oracle.SubRingConsts(N, q) for q in mods]
        xa = [ka[0, l].cpu().numpy().view(np.uint64) for l in range(L)]
        xb = [kb[0, l].cpu().numpy().view(np.uint64) for l in range(L)]
        z = np.zeros(N, dtype=np.uint64)
        reps, t0 = 0, time.perf_counter()
        while reps < 1 or time.perf_counter() - t0 < min(args.cpu_seconds, 10.0):
            for l in range(L):
                fa, fb = oracle.ntt(xa[l], srs[l]), oracle.ntt(xb[l], srs[l])
                oracle.intt(oracle.vec_op(MUL, oracle.vec_op(MFORM, fa, z, z, 0, 0, mods[l]), fb, z, 0, 0, mods[l]), srs[l])
            reps += 1
        out["cpu_baseline"] = {"value": reps / (time.perf_counter() - t0), "unit": "poly-mul/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                               "sample": "%d products of one %d-limb N=2^%d poly pair through the C restatement (oracle/ring_oracle.c), single thread" % (reps, L, logn)}
    print(json.dumps(out), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ workload: ctmul (BASELINE config 4)
def run_ctmul(args):
    """a step = matrix_ckks.Evaluator.Mul (schemes/matrix_ckks/evaluator.go:114-192) on `--batch` pairs of degree-1 ciphertexts over the 3N ring
    N = 3*2^16 (`--logn 14`: the other reading, 3*2^14), 24 limbs: 4 forward 3N transforms, the tensoring, 3 inverse transforms; batch-sharded
    (BASELINE: 1024 pairs over 8 GPUs = 128 per GPU, the default batch), no data-path collective"""
    import numpy as np
    import torch
    import matrix_fhe_lattigo_amd as rh
    sys.path.insert(0, os.path.join(ROOT, "tools"))
    from primes3n import moduli_3n
    rank, local_rank, world, dist, dev, ranks_seen = init_dist(args)
    N, L = 3 << args.logn, (args.limbs if any(a.startswith("--limbs") for a in sys.argv) else 24)
    B = args.batch if args.batch > 0 else 128
    mods = moduli_3n(N, L)
    ring = rh.Ring(N, mods, kind=rh.Matrix3N, device=local_rank)
    stream = torch.cuda.current_stream()
    ring.set_stream(stream.cuda_stream)
    for kv in args.tune:
        k, v = kv.split("=")
        ring.set_tuning(k, int(v))
    om = [int(w) for w in ring.constants()["omega3n"]]
    g = torch.Generator(device=dev); g.manual_seed(0xc4 + rank)
    qs = torch.tensor(mods, dtype=torch.int64, device=dev).view(1, L, 1)

    def mk(rand=True):
        t = torch.empty((B, L, N), dtype=torch.int64, device=dev)
        for b0 in range(0, B, 16):
            n = min(16, B - b0)
            t[b0:b0 + n] = (torch.randint(0, 1 << 62, (n, L, N), dtype=torch.int64, device=dev, generator=g) % qs) if rand else 0
        return t
    tin = [mk() for _ in range(4)]
    tout = [mk(False) for _ in range(3)]
    pin = [rh.DevicePoly.from_torch(ring, t) for t in tin]
    ct0, ct1 = rh.Ciphertext(pin[0:2]), rh.Ciphertext(pin[2:4])
    out = rh.Ciphertext([rh.DevicePoly.from_torch(ring, t) for t in tout])
    ev = rh.MatrixCKKSEvaluator(ring)                           # the default evaluator: block-order device NTT domain, carried as per-block tags

    def step():
        # Mul transforms coefficient-domain inputs IN PLACE (evaluator.go:136-149); the next step reads what it left as coefficient-domain
        # input again (canonical residues either way), so every step is the full 4 NTT + tensoring + 3 INTT
        ct0.IsNTT = ct1.IsNTT = False
        ev.Mul(ct0, ct1, out)

    wall, dev_ms, per_rank = timed_region(step, args, dist, dev, stream)
    ring.sync()
    verified = None
    spots = sorted({(0, 0), (B - 1, L - 1)})
    if not args.no_verify:
        # one more step of the same launch sequence on the same buffers, its inputs snapshotted first (raw device rows: what the step reads)
        import oracle
        snap = {s_: [t[s_[0], s_[1]].cpu().numpy().view(np.uint64).copy() for t in tin] for s_ in spots}
        step(); ring.sync()
        MUL, MULADD = rh.OPS["MUL_MONT"], rh.OPS["MUL_MONT_THEN_ADD"]
        z = np.zeros(N, dtype=np.uint64)
        ok = True
        for (p_, l), xs in snap.items():
            q, w = mods[l], om[l]
            A0, A1, B0, B1 = (oracle.ntt3n_forward(x, q, w) for x in xs)
            e = [oracle.vec_op(MUL, A0, B0, z, 0, 0, q),
                 oracle.vec_op(MULADD, A1, B0, oracle.vec_op(MUL, A0, B1, z, 0, 0, q), 0, 0, q),
                 oracle.vec_op(MUL, A1, B1, z, 0, 0, q)]
            for c in range(3):
                if not np.array_equal(tout[c][p_, l].cpu().numpy().view(np.uint64), oracle.ntt3n_backward(e[c], q, w)):
                    ok = False
        if not ok:
            sys.stderr.write("bench.py: rank %d: ct x ct Mul MISMATCH vs oracle\n" % rank)
        verified = all_ranks_ok(ok, args, dist, dev)
    if rank != 0:
        if dist is not None:
            dist.destroy_process_group()
        return
    dev_step_ms = dev_ms / args.steps
    value = world * B * args.steps / wall
    alg_bytes = (7 * 16.0 + 3 * 24.0 + 32.0) * N * L * B       # 7 transforms (16 N per limb), 3 MulCoeffsMontgomery (24 N), 1 ...ThenAdd (32 N): SURVEY 8(d)
    achieved = alg_bytes / (dev_step_ms * 1e-3) / 1e9
    res = {
        "metric": "ct x ct Mul/s on the 3N ring N=%d, %d RNS limbs (matrix_ckks.Evaluator.Mul, BASELINE config 4)" % (N, L),
        "value": value, "unit": "ct-mul/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
        "ms_per_step": wall * 1e3 / args.steps, "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
        "dtype": "u64", "data": "synthetic", "verified": verified,
        "config": {"workload": "matrix_ckks.Evaluator.Mul, N=3*2^%d, %d limbs (the first %d primes = 1 mod 3N above 2^60), batch %d ciphertext pairs/GPU, device-resident, "
                               "default evaluator (block-order NTT domain)" % (args.logn, L, L, B),
                   "parallelism": "batch-shard x%d, no data-path collective" % world,
                   "dist_backend": args.dist_backend if (world > 1 or args.force_dist) else None, "rccl_ranks" if args.dist_backend == "nccl" else "dist_ranks": ranks_seen,
                   "per_rank_device_ms_per_step": per_rank,
                   "verified_on": "every rank: one more step of the same sequence on the same buffers right after timing, %d spot (pair, limb) rows of its 3 outputs vs the oracle's transforms and products" % len(spots)},
        "roofline": {"bound": "hbm", "limited_by": "power/valu", "achieved": achieved, "peak": HBM_PEAK_GBS, "unit": "GB/s", "frac": achieved / HBM_PEAK_GBS, "traffic": None,
                     "algorithmic_bytes_per_step": alg_bytes, "device_ms_per_step": dev_step_ms,
                     "note": "algorithmic bytes = the reference's sequence: 7 transforms x 16 N + 3 x 24 N + 32 N per limb (SURVEY 8(d))"},
    }
    if world == 1 and not args.no_cpu:
        import oracle
        MUL, MULADD = rh.OPS["MUL_MONT"], rh.OPS["MUL_MONT_THEN_ADD"]
        xs = [t[0, 0].cpu().numpy().view(np.uint64) for t in tin]
        z = np.zeros(N, dtype=np.uint64)
        q, w = mods[0], om[0]
        t0 = time.perf_counter()
        A0, A1, B0, B1 = (oracle.ntt3n_forward(x, q, w) for x in xs)
        for e in (oracle.vec_op(MUL, A0, B0, z, 0, 0, q), oracle.vec_op(MULADD, A1, B0, oracle.vec_op(MUL, A0, B1, z, 0, 0, q), 0, 0, q), oracle.vec_op(MUL, A1, B1, z, 0, 0, q)):
            oracle.ntt3n_backward(e, q, w)
        t_limb = time.perf_counter() - t0
        res["cpu_baseline"] = {"value": 1.0 / (t_limb * L), "unit": "ct-mul/s", "cores": 1, "kind": "port", "cpu": cpu_model(),
                               "sample": "ONE of the %d limbs of one ciphertext pair timed (%.2f s: 7 transforms by the oracle's O(N log N) restatement + the 4 products), "
                                         "scaled by %d; single thread (the reference's own 3N transform is O(N^2) big-integer Horner: not runnable at this size)" % (L, t_limb, L)}
    print(json.dumps(res), flush=True)
    if dist is not None:
        dist.destroy_process_group()


# ------------------------------------------------------------------------------------------------ workload: keyswitch
def run_keyswitch(args):
    import numpy as np
    import torch
    import matrix_fhe_lattigo_amd as rh
    from matrix_fhe_lattigo_amd import sharding
    rank, local_rank, world, dist, dev, ranks_seen = init_dist(args)
    N = 1 << args.logn
    B = args.batch if args.batch > 0 else 64
    Q, P = QI60[:24], PI60[:6]
    nq, npl = len(Q), len(P)
    limb = args.shard == "limb"
    stream = torch.cuda.current_stream()
    ks = be = None
    if limb:
        ks = sharding.LimbShardedKeySwitch(N, Q, P, rank, world, dist=dist if world > 1 else None, device=local_rank)
        beta, ownQ, ownP = ks.beta, ks.ownQ, ks.ownP
        lo, hi = 0, B                                                   # every rank works on the whole batch, on its limbs
    else:
        ringQ, ringP = rh.Ring(N, Q, device=local_rank), rh.Ring(N, P, device=local_rank)
        ringQ.set_stream(stream.cuda_stream); ringP.set_stream(stream.cuda_stream)
        be = rh.BasisExtender(ringQ, ringP)
        beta, ownQ, ownP = (nq - 1 + npl) // npl, list(range(nq)), list(range(npl))
        lo, hi = sharding.poly_shard(B, rank, world)                   # this rank's polys, all limbs, the whole key
    nb = hi - lo
    # synthetic inputs: the SAME (seeded) full-size case on every rank, each keeps its limbs / polys (uniform key: SURVEY 8d)
    g = torch.Generator(device=dev)

    def limb_rows(seed, nlead, mods, i):
        """(nlead, N) uniform residues of global limb i: the unit of generation, identical on whichever rank draws it"""
        g.manual_seed(seed * 1000 + i)
        return torch.randint(0, 1 << 62, (nlead, N), dtype=torch.int64, device=dev, generator=g) % mods[i]

    def rows(seed, lead, mods, own, sl=None):
        n0 = 1
        for d in lead:
            n0 *= d
        sel = slice(0, n0) if sl is None else sl            # sl: a range of the (single) leading axis -- this rank's polys
        shape = tuple(lead) if sl is None else (sel.stop - sel.start,)
        out = torch.empty(shape + (len(own), N), dtype=torch.int64, device=dev)
        flat = out.view(-1, len(own), N)
        for k, i in enumerate(own):
            flat[:, k] = limb_rows(seed, n0, mods, i)[sel]
        return out
    cx = rows(1, (B,), Q, ownQ, slice(lo, hi))
    evkQ = rows(2, (beta, 2), Q, ownQ)
    evkP = rows(3, (beta, 2), P, ownP) if ownP else None
    ct0, ct1 = torch.empty_like(cx), torch.empty_like(cx)
    if limb:
        def step():
            ks.GadgetProduct(cx, evkQ, evkP, ct0, ct1, chunks=args.chunks)
    else:
        dp = lambda ring, t: rh.DevicePoly.from_torch(ring, t.view(-1, t.shape[-2], N))
        pcx, p0, p1 = dp(ringQ, cx), dp(ringQ, ct0), dp(ringQ, ct1)

        def step():
            if nb:
                be.GadgetProduct(nq - 1, npl - 1, pcx, evkQ.data_ptr(), evkP.data_ptr(), beta, p0, p1)

    wall, dev_ms, per_rank = timed_region(step, args, dist, dev, stream)
    exch = {"exchanges_per_product": ks.exchanges, "bytes_received_per_gpu_per_product": ks.exchange_words * 8} if limb else None
    verified = None
    if not args.no_verify:                                 # EVERY rank: first, middle and last poly of its share against the oracle composition
        from oracle import compose
        host = lambda t: t.cpu().numpy().view(np.uint64)
        ekq = np.stack([host(limb_rows(2, beta * 2, Q, i)) for i in range(nq)], axis=1).reshape(beta, 2, nq, N)
        ekp = np.stack([host(limb_rows(3, beta * 2, P, j)) for j in range(npl)], axis=1).reshape(beta, 2, npl, N)
        ok = True
        for k in sorted({0, nb // 2, nb - 1}) if nb else []:
            cx_f = np.stack([host(limb_rows(1, B, Q, i)[lo + k]) for i in range(nq)])                          # poly lo + k, every limb
            e0, e1 = compose.gadget_product(N, Q, P, nq - 1, npl - 1, cx_f, ekq, ekp)
            ok = ok and np.array_equal(host(ct0[k]), e0[ownQ]) and np.array_equal(host(ct1[k]), e1[ownQ])
        if not ok:
            sys.stderr.write("bench.py: rank %d: key-switch OUTPUT MISMATCH vs the oracle composition\n" % rank)
        verified = all_ranks_ok(ok, args, dist, dev)
    if rank == 0:
        ms_per_step = wall * 1e3 / args.steps
        value = B * args.steps / wall
        launch_ms = dev_ms / args.steps
        # limb transforms x 16 N + the evaluation-key read 2 beta (L+k) 8 N, whole job
        limb_ntts = nq + beta * (nq + npl) - nq + 2 * (npl + nq)      # INTT(cx) + per digit NTTs (own limbs skipped: -nq in total) + 2 ModDowns
        alg = B * limb_ntts * 16.0 * N + 2.0 * beta * (nq + npl) * 8.0 * N * (1 if limb else world)   # batch-shard: every rank reads the whole key
        achieved = alg / (launch_ms * 1e-3) / 1e9
        par = ("limb-shard x%d (round-robin over Q++P): orchestration behind the C ABI (rh_kshard_gadget_product), %s chunks on two streams, two "
               "all-gathers per chunk (source limbs, P parts)" % (world, args.chunks or "auto")) if limb else \
              "batch-shard x%d: %s polys per rank, the 120 MiB key replicated, no data-path collective" % (world, "/".join(str(sharding.poly_shard(B, r, world)[1] - sharding.poly_shard(B, r, world)[0]) for r in range(world)))
        out = {
            "metric": "key-switch (hybrid gadget product)/s at N=2^%d, %d Q + %d P moduli, %s-sharded" % (args.logn, nq, npl, args.shard),
            "value": value, "unit": "key-switch/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": ms_per_step, "higher_is_better": True, "scaling": "strong", "vs_baseline": None,
            "dtype": "u64", "data": "synthetic", "verified": verified,
            "config": {"workload": "rlwe.Evaluator.GadgetProduct, N=2^%d, Q=Qi60[0:24], P=Pi60[0:6], beta=%d, batch %d polys per step (whole job), "
                                   "uniform key shared by the batch" % (args.logn, beta, B),
                       "parallelism": par, "shard": args.shard, "exchange": exch,
                       "dist_backend": args.dist_backend if world > 1 else None, "rccl_ranks" if args.dist_backend == "nccl" else "dist_ranks": ranks_seen,
                       "per_rank_device_ms_per_step": per_rank, "verified_on": "every rank (first, middle, last poly of its share, owned limbs)"},
            "roofline": {"bound": "hbm", "achieved": achieved, "peak": HBM_PEAK_GBS * world, "unit": "GB/s", "frac": achieved / (HBM_PEAK_GBS * world),
                         "traffic": None, "kernel": "whole gadget product (%d limb transforms + key multiply-accumulate + basis extensions per poly)" % limb_ntts,
                         "algorithmic_bytes_per_step": alg, "device_ms_per_step": launch_ms},
        }
        print(json.dumps(out), flush=True)
    if ks is not None:
        ks.close()
    if be is not None:
        be.close(); ringQ.close(); ringP.close()
    if dist is not None:
        dist.destroy_process_group()


def main():
    args = parse()
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        return spawn_ranks(args)
    if args.workload == "polymul":
        return run_polymul(args)
    if args.workload == "ctmul":
        return run_ctmul(args)
    if args.workload == "keyswitch":
        return run_keyswitch(args)
    return run_ntt(args)


if __name__ == "__main__":
    main()
