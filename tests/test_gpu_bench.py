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
