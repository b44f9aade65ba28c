#!/usr/bin/env python3
"""rocprofv3 --kernel-trace CSV -> the timeline of the LAST `n` library kernels (start offset, duration in microseconds, grid): trace_table.py <kernel_trace.csv> [n]"""
import csv, sys
rows = [r for r in csv.DictReader(open(sys.argv[1])) if "at::" not in r["Kernel_Name"] and "rocclr" not in r["Kernel_Name"]]
rows.sort(key=lambda r: int(r["Start_Timestamp"]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 20
last = rows[-n:]
t0 = int(last[0]["Start_Timestamp"])
print("%-66s %9s %9s %10s" % ("kernel", "start_us", "dur_us", "grid_x"))
for r in last:
    print("%-66s %9.1f %9.1f %10s" % (r["Kernel_Name"].split("(")[0][:66], (int(r["Start_Timestamp"]) - t0) / 1e3, (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3, r.get("Grid_Size_X", "")))
print("span of these %d kernels: %.1f us" % (len(last), (int(last[-1]["End_Timestamp"]) - t0) / 1e3))
