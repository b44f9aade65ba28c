# GPU box: where does the forward tile kernel lose against the register-resident butterfly rate?  TIMING ONLY (variants compute garbage).
cd $GRAFT_REPO_ROOT
run() { python bench.py --no-cpu --no-verify "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read().strip().splitlines()[-1]); print(round(d['ms_per_step'],3), {k: round(v,3) for k,v in d['roofline']['standalone_kernel_ms'].items()})"; }
echo "default"; run
for e in ${EXPS:-1 2 4 8 7}; do echo "exp$e"; RINGHIP_LIB=build/variants/libringhip_exp$e.so run; done
