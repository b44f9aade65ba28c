#!/bin/bash
# fabric traffic of the one-pass team kernel (FETCH_SIZE / WRITE_SIZE in separate passes, as the guide prescribes)
cd /tmp && export TMPDIR=/tmp && cd "$GRAFT_REPO_ROOT"
O=gpurun_out/team_pmc; rm -rf $O; mkdir -p $O
S=${1:-9}; W=${2:-4}
rocprofv3 --kernel-trace --pmc FETCH_SIZE -d $O/f -o f --output-format csv -- python3 experiments/team_one_pass/exp_team.py one $S $W > $O/f.log 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE -d $O/w -o w --output-format csv -- python3 experiments/team_one_pass/exp_team.py one $S $W > $O/w.log 2>&1
python3 - <<PY
import csv, collections
for tag, path in (("FETCH_SIZE", "$O/f/f_counter_collection.csv"), ("WRITE_SIZE", "$O/w/w_counter_collection.csv")):
    tot = collections.defaultdict(lambda: [0.0, 0])
    for row in csv.DictReader(open(path)):
        if "ntt_fwd_team" in row["Kernel_Name"]:
            tot[row["Counter_Name"]][0] += float(row["Counter_Value"]); tot[row["Counter_Name"]][1] += 1
    for k, (v, n) in tot.items():
        print(tag, k, "sum over %d rows" % n, v, "-> per launch (KiB units, /5 launches):", v / 5)
PY
tail -2 $O/f.log
