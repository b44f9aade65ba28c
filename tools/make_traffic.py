#!/usr/bin/env python3
"""profiles/latest_traffic.json from the two rocprofv3 PMC passes of `bench.py` (FETCH_SIZE and WRITE_SIZE, separate runs).

usage: make_traffic.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> <batch> <launches_per_step>
gfx950 corrections as in tools/pmc_summary.py (KiB units, FETCH_SIZE x2).  The step's traffic = launches_per_step x the
average of the pipelined launch (ntt_fwd_fused_asm); Infinity-Cache hits are included in these counters."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_tree_hash          # ties the counters to the kernel sources they were collected on (bench.py drops them when it differs)


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"].split("(")[0].replace("void ", "")].append(float(r["Counter_Value"]))
    return agg


f, w = per_kernel(sys.argv[1], "FETCH_SIZE"), per_kernel(sys.argv[2], "WRITE_SIZE")
batch, launches = int(sys.argv[4]), int(sys.argv[5])
fused = [k for k in f if "ntt_fwd_fused" in k]
assert len(fused) == 1, fused
k = fused[0]
# the first and last launch of a step carry one item only; averaging over all launches of whole steps is exact for the step
fetch_step = sum(f[k]) / len(f[k]) * launches * 1024 * 2
write_step = sum(w[k]) / len(w[k]) * launches * 1024
out = {
    "what": "bench.py default config: bytes crossing the L2<->fabric boundary per step (one forward transform of the batch = %d %s launches)" % (launches, k),
    "method": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes; KiB units; FETCH_SIZE doubled per the gfx950 correction "
              "(calibrated on ntt_fwd_cols: raw 4.0 GiB for exactly 8 GiB read); Infinity-Cache hits are included in these counters",
    "fetch_bytes_per_step": fetch_step, "write_bytes_per_step": write_step, "hbm_bytes_per_step": fetch_step + write_step,
    "algorithmic_bytes_per_step": 16.0 * 65536 * 16 * batch, "launches_per_step": launches,
    "launches_sampled": {"fetch": len(f[k]), "write": len(w[k])},
    "standalone_kernels_per_launch": {n: {"fetch": sum(f[n]) / len(f[n]) * 2048, "write": sum(w.get(n, [0])) / max(len(w.get(n, [0])), 1) * 1024}
                                      for n in f if "ntt" in n and n != k},
    "config": {"logn": 16, "limbs": 16, "batch": batch},
    "hbm_bytes_per_poly": (fetch_step + write_step) / batch,
    "csrc_tree": csrc_tree_hash(),
}
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
