#!/bin/bash
# GPU box: regenerates everything under profiles/ that the bench line cites.  usage: tools/refresh_profiles.sh <tag>  (e.g. r01)
set -e
tag=${1:-r01}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
echo "[1/5] bench"; python3 bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/${tag}_bench.json
echo "[2/5] kernel trace"; rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu > $O/kt.log 2>&1
cp $O/kt/kt_kernel_stats.csv $O/${tag}_kernel_stats.csv
echo "[3/5] pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/pf.log 2>&1
echo "[4/5] pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w --output-format csv -- python3 bench.py --no-cpu --steps 3 --warmup 1 > $O/pw.log 2>&1
python3 tools/make_traffic.py $O/pf/f_counter_collection.csv $O/pw/w_counter_collection.csv $O/latest_traffic.json 1024 9 > /dev/null
# keep only the NTT rows of the raw counter files (the full files are large)
for x in f w; do d=$([ $x = f ] && echo pf || echo pw); head -1 $O/$d/${x}_counter_collection.csv > $O/${tag}_pmc_${x}.csv; grep ntt_ $O/$d/${x}_counter_collection.csv >> $O/${tag}_pmc_${x}.csv || true; done
echo "[5/5] ops"; python3 tools/bench_ops.py > $O/${tag}_bench_ops.json 2>$O/ops.err
ls -la $O | head -30
