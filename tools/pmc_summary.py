#!/usr/bin/env python3
"""Summarises rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE CSVs (separate passes) into per-kernel bytes per launch.

gfx950 corrections (/opt/skills/guides/MI355X_MICROARCH.md, HBM section): both counters are in KiB; FETCH_SIZE reports
exactly 1/2 of the bytes of a coalesced streaming read (checked here on ntt_fwd_cols, whose reads are exactly 8*N per
limb: the raw counter shows 4 GiB for 8 GiB read) -> doubled; WRITE_SIZE is exact.
usage: pmc_summary.py <fetch_counter_collection.csv> <write_counter_collection.csv> <out.json> [steps_kernel_names...]"""
import collections
import csv
import json
import sys


def per_kernel(path, counter):
    agg = collections.defaultdict(list)
    for r in csv.DictReader(open(path)):
        if r["Counter_Name"] == counter:
            agg[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: sum(v) / len(v) for k, v in agg.items()}, {k: len(v) for k, v in agg.items()}


fetch, nf = per_kernel(sys.argv[1], "FETCH_SIZE")
write, nw = per_kernel(sys.argv[2], "WRITE_SIZE")
out = {"units": "bytes per launch", "corrections": "FETCH_SIZE KiB x2 (gfx950), WRITE_SIZE KiB x1", "kernels": {}}
total = 0.0
for k in sorted(set(fetch) | set(write)):
    if not any(t in k for t in ("ntt", "vec_op", "bext", "gadget", "perm", "automorphism", "rescale")):
        continue
    f = fetch.get(k, 0.0) * 1024 * 2
    w = write.get(k, 0.0) * 1024
    short = k.split("(")[0].replace("void ", "")
    out["kernels"][short] = {"fetch_bytes": f, "write_bytes": w, "launches_sampled": nf.get(k, 0)}
    total += f + w
out["hbm_bytes_per_step"] = total
json.dump(out, open(sys.argv[3], "w"), indent=1)
print(json.dumps(out, indent=1))
