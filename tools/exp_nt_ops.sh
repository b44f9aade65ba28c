#!/bin/bash
# every other op with and without the non-temporal data streams (gpurun_in/libringhip_base.so = RH_ASM_DATA_FLAGS="" build), alternating runs
mkdir -p gpurun_out
for rep in 1 2; do
  RINGHIP_LIB=$PWD/gpurun_in/libringhip_base.so python tools/bench_ops.py > gpurun_out/ops_base$rep.json 2>/dev/null || exit 1
  python tools/bench_ops.py > gpurun_out/ops_nt$rep.json 2>/dev/null || exit 1
done
python - <<PY
import json
L = lambda n: json.load(open("gpurun_out/%s.json" % n))["results"]
a1, a2, b1, b2 = L("ops_base1"), L("ops_base2"), L("ops_nt1"), L("ops_nt2")
for x1, x2, y1, y2 in zip(a1, a2, b1, b2):
    a, b = (x1["ms"] + x2["ms"]) / 2, (y1["ms"] + y2["ms"]) / 2
    print("%-96s base %7.3f %7.3f | nt %7.3f %7.3f | %+.1f %%" % (y1["op"][:96], x1["ms"], x2["ms"], y1["ms"], y2["ms"], (b / a - 1) * 100))
PY
