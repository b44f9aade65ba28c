"""GPU: bench.py end to end at reduced size -- the JSON contract, the self-started N > 1 ranks (`--gpus 2` with no launcher:
two processes on this box's one GPU, gloo rendezvous on 127.0.0.1 = the rehearsal of the RCCL path), the limb-sharded
key-switch workload, and the oracle check bench.py makes of its own timed output (`verified`)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def run_bench(*extra, timeout=900):
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py")] + list(extra), capture_output=True, text=True, timeout=timeout, env=env)
    assert p.returncode == 0, p.stderr[-2000:]
    lines = [l for l in p.stdout.splitlines() if l.startswith("{")]
    assert len(lines) == 1, p.stdout
    return json.loads(lines[0])


def test_bench_one_gpu_line_and_self_check():
    j = run_bench("--batch", "260", "--steps", "3", "--warmup", "1", "--cpu-seconds", "1")
    for k in ("metric", "value", "unit", "n_gpus", "steps", "warmup", "ms_per_step", "higher_is_better", "scaling", "vs_baseline", "dtype", "data",
              "config", "roofline", "cpu_baseline", "verified"):
        assert k in j, k
    assert j["n_gpus"] == 1 and j["steps"] == 3 and j["verified"] is True and j["vs_baseline"] is None
    r = j["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and r["launches_per_step"] == 4
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["cores"] == 1 and j["cpu_baseline"]["value"] > 0


def test_bench_gpus_2_starts_two_ranks():
    # the driver's N > 1 form without torchrun: the parent spawns the ranks; here both share cuda:0 and rendezvous over gloo
    j = run_bench("--gpus", "2", "--single-device", "--dist-backend", "gloo", "--batch", "8", "--steps", "2", "--warmup", "1", "--no-cpu")
    assert j["n_gpus"] == 2 and j["config"]["dist_ranks"] == 2 and j["verified"] is True
    assert j["scaling"] == "weak" and "x2" in j["config"]["parallelism"]


def test_bench_keyswitch_workload_two_ranks():
    j = run_bench("--workload", "keyswitch", "--gpus", "2", "--single-device", "--dist-backend", "gloo", "--batch", "2", "--steps", "1", "--warmup", "1",
                  "--logn", "13")
    assert j["n_gpus"] == 2 and j["unit"] == "key-switch/s" and j["verified"] is True and j["scaling"] == "strong"
    assert j["config"]["dist_ranks"] == 2


def test_bench_polymul_workload_one_gpu_and_two_ranks():
    # BASELINE config 3 through bench.py: the JSON contract with roofline + cpu_baseline at N = 1; two ranks batch-sharded, every rank verifies
    j = run_bench("--workload", "polymul", "--batch", "40", "--steps", "2", "--warmup", "1", "--cpu-seconds", "1")
    assert j["unit"] == "poly-mul/s" and j["n_gpus"] == 1 and j["verified"] is True and j["scaling"] == "weak" and "2^15" in j["metric"]
    r = j["roofline"]
    assert r["bound"] == "hbm" and abs(r["frac"] - r["achieved"] / r["peak"]) < 1e-12 and 0 < r["frac_vs_fused_lower_bound_24NL"] < r["frac"]
    assert j["cpu_baseline"]["kind"] == "port" and j["cpu_baseline"]["value"] > 0
    j = run_bench("--workload", "polymul", "--gpus", "2", "--single-device", "--dist-backend", "gloo", "--batch", "6", "--steps", "1", "--warmup", "1", "--logn", "13", "--no-cpu")
    assert j["n_gpus"] == 2 and j["config"]["dist_ranks"] == 2 and j["verified"] is True and "2^13" in j["metric"]


def test_bench_ctmul_workload_one_gpu_and_two_ranks():
    # BASELINE config 4 through bench.py at a reduced ring (N = 3*2^13, 3 limbs): contract fields, verification on every rank
    j = run_bench("--workload", "ctmul", "--logn", "13", "--limbs", "3", "--batch", "3", "--steps", "2", "--warmup", "1")
    assert j["unit"] == "ct-mul/s" and j["n_gpus"] == 1 and j["verified"] is True and "N=24576" in j["metric"]
    assert j["roofline"]["bound"] == "hbm" and j["cpu_baseline"]["value"] > 0
    j = run_bench("--workload", "ctmul", "--logn", "13", "--limbs", "2", "--batch", "2", "--steps", "1", "--warmup", "1", "--gpus", "2", "--single-device",
                  "--dist-backend", "gloo", "--no-cpu")
    assert j["n_gpus"] == 2 and j["config"]["dist_ranks"] == 2 and j["verified"] is True


# ---- pre-flight of the first real 8-GPU run: the driver's N > 1 form rehearsed with MANY ranks on this box's one GPU over
# gloo.  The GPU box's process guard allows 6 processes on the card at once and this pytest process already holds a context, so the rehearsal
# runs 5 ranks there (RH_BENCH_REHEARSE_RANKS overrides where no guard applies); everything that depends on the rank count -- rendezvous, shards
# with remainders (7 polys over 5 ranks, 30 limbs over 5 ranks), per-rank statistics, verification on every rank, the chunked exchanges --
# is exercised the same way, and the 8-rank rendezvous / gather / shard arithmetic itself runs on the CPU in tests/test_multiprocess.py.
REHEARSE = int(os.environ.get("RH_BENCH_REHEARSE_RANKS", "5"))


def test_bench_ntt_many_ranks_rehearsal_with_final_gather():
    n = REHEARSE
    j = run_bench("--gpus", str(n), "--single-device", "--dist-backend", "gloo", "--batch", "4", "--steps", "2", "--warmup", "1", "--no-cpu", "--logn", "14",
                  "--gather", "--gather-polys", "2")
    assert j["n_gpus"] == n and j["config"]["dist_ranks"] == n and j["verified"] is True
    pr = j["config"]["per_rank_device_ms_per_step"]
    assert len(pr["device_ms_per_step"]) == n and 0 < pr["min"] <= pr["max"]
    g = j["final_gather"]
    assert g["verified"] is True and g["polys_per_rank"] == 2 and g["bytes_received_per_gpu"] == (n - 1) * 2 * 16 * (1 << 14) * 8 and g["ms"] > 0
    assert abs(j["value"] - n * 4 * 2 / (j["ms_per_step"] * 2e-3)) < 1e-6 * j["value"]          # value = the units ALL ranks processed / the max-over-ranks time


@pytest.mark.parametrize("shard", ["limb", "batch"])
def test_bench_keyswitch_many_ranks_rehearsal(shard):
    n = REHEARSE
    j = run_bench("--workload", "keyswitch", "--shard", shard, "--gpus", str(n), "--single-device", "--dist-backend", "gloo", "--batch", "7", "--steps", "1",
                  "--warmup", "1", "--logn", "13")
    assert j["n_gpus"] == n and j["config"]["dist_ranks"] == n and j["verified"] is True and j["config"]["shard"] == shard
    assert len(j["config"]["per_rank_device_ms_per_step"]["device_ms_per_step"]) == n
    if shard == "limb":
        ex = j["config"]["exchange"]
        assert ex["exchanges_per_product"] == 2 * 4 and ex["bytes_received_per_gpu_per_product"] > 0       # 7 polys: chunks of 2, 2, 2, 1
    else:
        assert j["config"]["exchange"] is None and "no data-path collective" in j["config"]["parallelism"]


def test_bench_parent_fails_fast_when_a_rank_dies():
    # a rank killed mid-run must not leave the parent waiting in a collective: it stops the others and exits non-zero well inside the time-out
    import signal
    import time
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.Popen([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--single-device", "--dist-backend", "gloo", "--batch", "64", "--steps",
                          "2000000", "--warmup", "1", "--no-cpu", "--no-verify", "--logn", "14", "--rank-timeout", "600"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
                         text=True, env=env)
    victim, deadline = None, time.time() + 240
    while victim is None and time.time() < deadline:                     # a direct child of the parent that is a rank (its command line is bench.py's)
        time.sleep(2.0)
        for d in os.listdir("/proc"):
            if not d.isdigit():
                continue
            try:
                st = open("/proc/%s/stat" % d).read().rsplit(")", 1)[1].split()
                if int(st[1]) == p.pid and "bench.py" in open("/proc/%s/cmdline" % d).read():
                    victim = int(d)
            except (OSError, IndexError, ValueError):
                continue
    def rank_pids():
        out = []
        for d in os.listdir("/proc"):
            if d.isdigit():
                try:
                    st = open("/proc/%s/stat" % d).read().rsplit(")", 1)[1].split()
                    if int(st[1]) == p.pid and "bench.py" in open("/proc/%s/cmdline" % d).read():
                        out.append(int(d))
                except (OSError, IndexError, ValueError):
                    pass
        return out
    try:
        assert victim is not None, "no rank process found"
        time.sleep(20.0)                                                   # let the ranks get into their timed loop
        # the exact PID of one rank, and SIGTERM, not SIGKILL: the rank leaves at its next step boundary with its stream drained (bench.py:
        # _leave_if_stopped) -- for the parent and the other ranks a rank that died mid-run all the same, but no process is torn down with kernels in
        # flight on the GPU this pytest process shares with it
        os.kill(victim, signal.SIGTERM)
        t0 = time.time()
        try:
            _out, err = p.communicate(timeout=120)
        except subprocess.TimeoutExpired:
            raise AssertionError("parent still waiting 120 s after a rank died")
        assert p.returncode != 0 and "rank exit codes" in err and time.time() - t0 < 120
    finally:                                                               # whatever happened: no rank of this test outlives it (exact PIDs, never a pattern)
        left = rank_pids()
        for pid in left:                                                   # first the polite way (drain, then exit) ...
            try:
                os.kill(pid, signal.SIGTERM)
            except OSError:
                pass
        t1 = time.time()
        while left and time.time() - t1 < 30:
            time.sleep(0.5)
            left = rank_pids()
        for pid in left:                                                   # ... SIGKILL only for what is left (blocked in a host-side collective: GPU idle)
            try:
                os.kill(pid, signal.SIGKILL)
            except OSError:
                pass
        if p.poll() is None:
            p.kill()


def test_bench_drops_stale_counters(tmp_path):
    # profiles/latest_*.json are tied to the kernel sources by a tree hash: a line printed from other sources carries null, not old counters
    sys.path.insert(0, ROOT)
    import bench
    tree = bench.csrc_tree_hash()
    j = run_bench("--batch", "130", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-power")
    r = j["roofline"]
    assert r["csrc_tree"] == tree and r["limited_by"] == "power/valu"
    tj = bench.load_json("latest_traffic.json")
    if tj and tj.get("csrc_tree") == tree:
        assert r["traffic"] is not None and r["traffic_note"] is None
    else:
        assert r["traffic"] is None and "refresh" in r["traffic_note"]


def test_bench_rccl_calls_with_a_world_of_one():
    # every collective the N > 1 run makes -- all-reduce of ones, barriers, all-gather of the per-rank clocks, AND of the verification flags, the
    # final gather as ONE all_gather_into_tensor on device memory -- through the REAL RCCL backend, with one rank on the box's one GPU
    j = run_bench("--force-dist", "--dist-backend", "nccl", "--batch", "130", "--steps", "2", "--warmup", "1", "--no-cpu", "--no-power", "--gather", "--gather-polys", "4")
    assert j["n_gpus"] == 1 and j["config"]["dist_backend"] == "nccl" and j["config"]["rccl_ranks"] == 1 and j["verified"] is True
    g = j["final_gather"]
    assert g["verified"] is True and "RCCL" in g["what"] and g["bytes_received_per_gpu"] == 0


def test_sharded_key_switch_callback_over_rccl_with_a_world_of_one(tmp_path):
    # the all-gather thunk the library calls back (sharding.LimbShardedKeySwitch._allgather), with the nccl backend and a SIDE stream: a world of
    # one makes RCCL copy the block onto itself, which is exactly the call a node makes (ExternalStream of the library's stream, views of the arena)
    code = r'''
import os, sys, socket
sys.path.insert(0, %r)
s = socket.socket(); s.bind(("127.0.0.1", 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK="0", WORLD_SIZE="1")
import torch, torch.distributed as dist
dev = torch.device("cuda", 0); torch.cuda.set_device(0)
dist.init_process_group("nccl", device_id=dev)
import matrix_fhe_lattigo_amd as rh
from matrix_fhe_lattigo_amd import sharding
from bench import QI60, PI60
ks = sharding.LimbShardedKeySwitch(4096, QI60[:4], PI60[:2], 0, 1, dist=dist)
words = 3 * 4096
ks._arena = torch.zeros(4 * words, dtype=torch.int64, device=dev); ks._arena_words = 4 * words
ks._arena[:words] = torch.arange(words, dtype=torch.int64, device=dev)
side = torch.cuda.Stream(device=dev)
torch.cuda.synchronize()
base = ks._arena.data_ptr()
rc = ks._allgather(None, base, base + 2 * words * 8, words, side.cuda_stream)
side.synchronize()
assert rc == 0, ks.cb_error
assert torch.equal(ks._arena[2 * words:3 * words], ks._arena[:words]) and ks.exchanges == 1
rc = ks._allgather(None, base, base + 3 * words * 8, words, None)          # the NULL stream: torch's default stream
torch.cuda.synchronize()
assert rc == 0 and torch.equal(ks._arena[3 * words:], ks._arena[:words])
ks.close(); dist.destroy_process_group()
print("RCCL_CALLBACK_OK")
''' % ROOT
    f = tmp_path / "cb.py"
    f.write_text(code)
    env = {k: v for k, v in os.environ.items() if k not in ("RANK", "LOCAL_RANK", "WORLD_SIZE", "MASTER_PORT")}
    p = subprocess.run([sys.executable, str(f)], capture_output=True, text=True, timeout=300, env=env)
    assert p.returncode == 0 and "RCCL_CALLBACK_OK" in p.stdout, p.stdout[-1500:] + p.stderr[-3000:]
