set -o pipefail
python -m pytest tests -m gpu -x -q > gpurun_out/r03_t7.log 2>&1; echo "pytest rc=$?" >> gpurun_out/r03_t7.log; tail -3 gpurun_out/r03_t7.log | cut -c1-300
python tools/bench_sizes.py > gpurun_out/r03_sizes.json 2> gpurun_out/r03_sizes.err; tail -6 gpurun_out/r03_sizes.err | cut -c1-400
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
for lg in 13 14; do
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d gpurun_out/op_f$lg -o f --output-format csv -- python3 tools/bench_sizes.py logn=$lg > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d gpurun_out/op_w$lg -o w --output-format csv -- python3 tools/bench_sizes.py logn=$lg > /dev/null 2>&1
python3 tools/pmc_summary.py gpurun_out/op_f$lg/f_counter_collection.csv gpurun_out/op_w$lg/w_counter_collection.csv gpurun_out/r03_onepass_pmc_traffic_logn$lg.json | tail -12
rm -rf gpurun_out/op_f$lg gpurun_out/op_w$lg
done
bash tools/refresh_profiles.sh r03 > gpurun_out/refresh.log 2>&1; tail -3 gpurun_out/refresh.log
