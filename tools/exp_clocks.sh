#!/bin/bash
# what the chip's clocks and power do under the headline kernel (vs idle)
mkdir -p gpurun_out
out=gpurun_out/exp_clocks.txt
: > $out
echo "== idle" >> $out
rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|socclk|Power|power" >> $out
python bench.py --no-cpu --no-verify --steps 1500 --warmup 3 > gpurun_out/exp_clocks_bench.json 2>/dev/null &
pid=$!
sleep 6
for i in 1 2 3 4 5 6; do
  echo "== under load sample $i" >> $out
  rocm-smi --showclocks --showpower 2>&1 | grep -E "sclk|mclk|fclk|Power|power" >> $out
  sleep 0.7
done
wait $pid
python -c "import json; d=json.loads(open('gpurun_out/exp_clocks_bench.json').readlines()[-1]); print('bench', d['value'], d['ms_per_step'])" >> $out
cat $out
