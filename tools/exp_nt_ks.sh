#!/bin/bash
# key switch (config 5 shapes) against the batch: default-policy library (gpurun_in/libringhip_base.so) vs the shipped one
for B in 1 8 32 64; do
  for rep in 1 2; do
    echo -n "base "; RINGHIP_LIB=$PWD/gpurun_in/libringhip_base.so python tools/bench_ks.py 2048 $B 2>/dev/null | tail -1
    echo -n "ship "; python tools/bench_ks.py 2048 $B 2>/dev/null | tail -1
  done
done
