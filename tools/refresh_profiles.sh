#!/bin/bash
# GPU box: regenerates everything under profiles/ that the bench line cites.  usage: tools/refresh_profiles.sh <tag>  (e.g. r02)
# One rocprofv3 run per counter group; --pmc runs carry --kernel-trace only (gpurun rule).  rocprofv3 gets the program itself after `--`.
set -e
tag=${1:-r03}
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh; rm -rf $O; mkdir -p $O
echo "[2/9] kernel trace"; rocprofv3 --kernel-trace --stats -d $O/kt -o kt --output-format csv -- python3 bench.py --no-cpu --no-verify --no-power > $O/kt.log 2>&1
cp $O/kt/kt_kernel_stats.csv $O/${tag}_kernel_stats.csv
echo "[3/9] pmc fetch"; rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/pf -o f --output-format csv -- python3 bench.py --no-cpu --no-verify --no-power --steps 3 --warmup 1 > $O/pf.log 2>&1
echo "[4/9] pmc write"; rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/pw -o w --output-format csv -- python3 bench.py --no-cpu --no-verify --no-power --steps 3 --warmup 1 > $O/pw.log 2>&1
python3 tools/make_traffic.py $O/pf/f_counter_collection.csv $O/pw/w_counter_collection.csv $O/latest_traffic.json 1024 9 > /dev/null
for x in f w; do d=$([ $x = f ] && echo pf || echo pw); n=$([ $x = f ] && echo fetch_size || echo write_size); head -1 $O/$d/${x}_counter_collection.csv > $O/${tag}_pmc_${n}.csv; grep ntt_ $O/$d/${x}_counter_collection.csv >> $O/${tag}_pmc_${n}.csv || true; done
echo "[5/9] pmc sq"; rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -d $O/ps -o s --output-format csv -- python3 bench.py --no-cpu --no-verify --no-power --steps 3 --warmup 1 > $O/ps.log 2>&1
python3 tools/make_valu.py $O/ps/s_counter_collection.csv $O/latest_valu.json 1024 9 > /dev/null
python3 tools/pmc_table.py $O/ps > $O/${tag}_pmc_sq_counters.txt
# the bench line LAST of the headline group: it reads profiles/latest_*.json, which must be the counters of THIS tree (csrc_tree stamp)
cp $O/latest_traffic.json $O/latest_valu.json profiles/
echo "[1/9] bench"; python3 bench.py > $O/bench.log 2>$O/bench.err; tail -1 $O/bench.log > $O/${tag}_bench.json
echo "[6/9] 3N kernels (config 4 ring, reference order and block order)"
for bo in 0 1; do
  rocprofv3 --kernel-trace --stats -d $O/k3_$bo -o k --output-format csv -- python3 tools/bench_3n.py 16 24 16 $bo > $O/k3_$bo.log 2>&1
  cp $O/k3_$bo/k_kernel_stats.csv $O/${tag}_3n_kernel_stats_order$bo.csv
  rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f3_$bo -o f --output-format csv -- python3 tools/bench_3n.py 16 24 16 $bo > /dev/null 2>&1
  rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w3_$bo -o w --output-format csv -- python3 tools/bench_3n.py 16 24 16 $bo > /dev/null 2>&1
  python3 tools/pmc_summary.py $O/f3_$bo/f_counter_collection.csv $O/w3_$bo/w_counter_collection.csv $O/${tag}_3n_pmc_traffic_order$bo.json > /dev/null
done
echo "[7/9] key switch"; rocprofv3 --kernel-trace --stats -d $O/ks -o ks --output-format csv -- python3 tools/bench_ks.py 2048 64 > $O/ks.log 2>&1
cp $O/ks/ks_kernel_stats.csv $O/${tag}_keyswitch_kernel_stats.csv
echo "[8/9] ops"; python3 tools/bench_ops.py > $O/${tag}_bench_ops.json 2>$O/ops.err
echo "[9/9] key-switch workload line"; python3 bench.py --workload keyswitch --no-cpu > $O/${tag}_bench_keyswitch.json 2>$O/bks.err || true
rm -rf $O/kt $O/pf $O/pw $O/ps $O/k3_* $O/f3_* $O/w3_* $O/ks
ls -la $O | head -40
