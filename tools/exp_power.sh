#!/bin/bash
# package power and shader clock under (a) pure VALU streams held for seconds (tools/micro_power.hip), (b) the kernels of the
# headline transform (tools/exp_power.py).  rocm-smi is sampled from this shell while the stream runs.
mkdir -p gpurun_out
out=gpurun_out/exp_power.txt
: > $out
for c in 0 1 2 3 4 5 6 7 8 9; do
  tools/micro_power.bin 4 $c 4 2>&1 | grep held >> $out &
  pid=$!
  sleep 2.2
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|Package Power" | tr '\n' ' ' >> $out
  echo >> $out
  wait $pid
done
python tools/exp_power.py >> $out 2>&1
cat $out
