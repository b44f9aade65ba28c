# rocprofv3 kernel stats of one command: tools/prof_one.sh <tag> <script> [args...]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT && mkdir -p gpurun_out/prof
tag=$1; shift
rocprofv3 --kernel-trace --stats -d gpurun_out/prof/$tag -o p --output-format csv -- python3 "$@" > gpurun_out/prof/$tag.log 2>&1
tail -1 gpurun_out/prof/$tag.log | cut -c1-200
cp gpurun_out/prof/$tag/p_kernel_stats.csv gpurun_out/prof/$tag.csv && rm -rf gpurun_out/prof/$tag
