#!/bin/bash
# headline step time against the pipelined span size (polys per fused launch; 16 rows per poly): does an Infinity-Cache-resident
# hand-over (span << 256 MiB) save energy under the 1400 W cap?
mkdir -p gpurun_out
out=gpurun_out/exp_span.txt
: > $out
for c in 128 64 32 16 8 4 256; do
  echo "chunk_polys=$c ($((c*8)) MiB per span)" >> $out
  python bench.py --no-cpu --no-verify --steps 30 --warmup 3 --chunk $c 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value'], d['ms_per_step'])" >> $out || exit 1
done
cat $out
