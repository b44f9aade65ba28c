#!/usr/bin/env python3
"""Per-kernel averages of every counter found in rocprofv3 counter_collection CSVs under a directory tree."""
import collections, csv, glob, sys
agg = collections.defaultdict(lambda: collections.defaultdict(list))
for path in glob.glob(sys.argv[1] + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(path)):
        k = r["Kernel_Name"].split("(")[0].replace("void ", "")
        if "ntt" not in k and "vec" not in k and "bext" not in k:
            continue
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k in sorted(agg):
    print(k)
    for c in sorted(agg[k]):
        v = agg[k][c]
        print(f"   {c:28s} {sum(v)/len(v):16.1f}   (n={len(v)})")
