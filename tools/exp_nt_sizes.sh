#!/bin/bash
# non-temporal data streams against the batch size (working set = batch x 8 MiB at N = 2^16, L = 16; Infinity Cache = 256 MiB)
mkdir -p gpurun_out
out=gpurun_out/exp_nt_sizes.txt; : > $out
for B in 2 4 8 16 32 64 128; do
  for lib in base nt base nt; do
    if [ $lib = base ]; then export RINGHIP_LIB=$PWD/gpurun_in/libringhip_base.so; else unset RINGHIP_LIB; fi
    python bench.py --no-cpu --no-verify --no-power --batch $B --steps 200 --warmup 20 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print('B=$B $lib', round(d['ms_per_step'],4))" >> $out || exit 1
  done
done
cat $out
