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
                          "200000", "--warmup", "1", "--no-cpu", "--no-verify", "--logn", "14", "--rank-timeout", "600"], stdout=subprocess.PIPE, stderr=subprocess.PIPE,
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
    assert victim is not None, "no rank process found"
    time.sleep(20.0)                                                       # let the ranks get into their timed loop
    os.kill(victim, signal.SIGKILL)                                        # the exact PID of one rank
    t0 = time.time()
    try:
        _out, err = p.communicate(timeout=120)
    except subprocess.TimeoutExpired:
        p.kill()
        raise AssertionError("parent still waiting 120 s after a rank died")
    assert p.returncode != 0 and "rank exit codes" in err and time.time() - t0 < 120


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
