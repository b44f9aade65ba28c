#!/bin/bash
# usage: tools/pmc_passes.sh <outdir> "<counters pass1>" "<counters pass2>" ... -- bench args (run on the GPU box)
# one rocprofv3 --pmc pass per counter group (kernel-trace only, per the gpurun rules)
out=$1; shift
groups=()
while [ "$1" != "--" ]; do groups+=("$1"); shift; done
shift
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT" || exit 1
i=0
for g in "${groups[@]}"; do
  rocprofv3 --kernel-trace --pmc $g -d "$out/p$i" -o p --output-format csv -- python3 bench.py --no-cpu "$@" > "$out.p$i.log" 2>&1 || { tail -5 "$out.p$i.log"; exit 1; }
  i=$((i+1))
done
