#!/bin/bash
# GPU box, round 3 additions to tools/refresh_profiles.sh: the sustained headline run, the host-pointer seam, the config 5 scaling inputs
# and both key-switch shardings rehearsed with 2 ranks on the one GPU.  usage: tools/refresh_profiles_extra.sh <tag>
set -e
tag=${1:-r03}
cd "$GRAFT_REPO_ROOT"
O=gpurun_out/refresh_extra; rm -rf $O; mkdir -p $O
echo "[1/6] sustained: 3000 consecutive in-place transforms of the 8 GiB batch"
python3 bench.py --steps 3000 --warmup 5 --no-cpu > $O/${tag}_bench_sustained.json 2>$O/sustained.err
echo "[2/6] host-pointer seam"; python3 tools/bench_host_path.py > $O/${tag}_host_path.json 2>$O/host.err
tools/micro_hostpath > $O/${tag}_micro_hostpath.txt 2>&1 || true
echo "[3/6] config 5 scaling inputs"; python3 tools/bench_kshard_pricing.py > $O/${tag}_kshard_pricing.json 2>$O/pricing.err
echo "[4/6] key switch, limb-shard, 2 ranks on one GPU (gloo)"
python3 bench.py --workload keyswitch --shard limb --gpus 2 --single-device --dist-backend gloo --steps 5 --warmup 2 > $O/${tag}_bench_keyswitch_limb_2ranks_one_gpu.json 2>$O/ksl.err
echo "[5/6] key switch, batch-shard, 2 ranks on one GPU (gloo)"
python3 bench.py --workload keyswitch --shard batch --gpus 2 --single-device --dist-backend gloo --steps 5 --warmup 2 > $O/${tag}_bench_keyswitch_batch_2ranks_one_gpu.json 2>$O/ksb.err
echo "[6/6] final gather leg, 2 ranks on one GPU (gloo, host-staged)"
python3 bench.py --gpus 2 --single-device --dist-backend gloo --batch 256 --gather --gather-polys 32 --steps 5 --warmup 2 --no-cpu > $O/${tag}_bench_gather_2ranks_one_gpu.json 2>$O/gather.err
ls -la $O
