cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-verify "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), round(d['value']))"; }
echo "default"; run; run
for p in 1 2 3; do echo "prio$p"; RINGHIP_LIB=build/variants/libringhip_prio$p.so run; done
for s in 512 1024 4096; do echo "span_rows $s"; run --tune auto_span_rows=$s; done
echo "default again"; run
