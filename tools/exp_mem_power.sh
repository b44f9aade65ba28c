#!/bin/bash
# package power under streaming traffic by access width (tools/micro_mem_power.hip): nJ per byte = (W - idle W) / (GB/s)
mkdir -p gpurun_out
out=gpurun_out/exp_mem_power.txt
: > $out
for m in 4 6; do
  tools/micro_mem_power.bin $m 4 >> $out 2>&1 &
  pid=$!
  sleep 2.4
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | tr '\n' ' ' >> $out
  echo >> $out
  wait $pid
done
cat $out
