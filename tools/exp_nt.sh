#!/bin/bash
# same-box A/B of the non-temporal data streams: tuning key nt_streams = 0 (default cache policy everywhere) against the library's default
# (profiles/r02_nt_policy_*.txt were produced with two builds of the library before the key existed: RINGHIP_LIB=<build without the _NT bodies>)
for rep in 1 2 3; do
  for v in 0 1; do
    echo -n "nt_streams=$v: "
    python bench.py --no-cpu --no-verify --no-power --steps 30 --tune nt_streams=$v "$@" 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d['value']), round(d['ms_per_step'],4))"
  done
done
