#!/bin/bash
# usage: tools/sweep.sh <flag> v1 v2 ... [-- extra bench args]   (GPU box): one bench.py run per value, prints value ms NTT/s
flag=$1; shift
vals=()
while [ $# -gt 0 ] && [ "$1" != "--" ]; do vals+=("$1"); shift; done
[ "$1" == "--" ] && shift
for v in "${vals[@]}"; do
  timeout -k 10 200 python bench.py --no-cpu $flag $v "$@" 2>/dev/null > /tmp/sweep.json || exit 1
  python - "$v" <<'PY' || exit 1
import json, sys
d = json.loads(open("/tmp/sweep.json").read().strip().splitlines()[-1])
print(sys.argv[1], round(d["ms_per_step"], 3), round(d["value"]))
PY
done
