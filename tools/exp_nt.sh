#!/bin/bash
# EXPERIMENT: cache-policy hints on the streams of the two-pass pipeline (generator env: RH_ASM_COLS_LOAD / RH_ASM_TILE_LOAD / RH_ASM_TILE_STORE)
mkdir -p gpurun_out
out=gpurun_out/exp_nt.txt; : > $out
for lib in base nt4 gnt gsc1nt gsysnt gsc1 base; do
  echo "$lib" >> $out
  for i in 1 2; do
    RINGHIP_LIB=$PWD/gpurun_in/libringhip_$lib.so python bench.py --no-cpu --no-verify --no-power --steps 30 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out || exit 1
  done
done
cat $out
