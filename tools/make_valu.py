#!/usr/bin/env python3
"""profiles/latest_valu.json: the VALU side of the headline kernel, read by bench.py for `roofline.valu`.

usage: make_valu.py <sq_counter_collection.csv> <out.json> <batch> <launches_per_step>
SQ_INSTS_VALU (wave-instructions) of the pipelined launch ntt_fwd_fused_asm, averaged over the sampled launches x launches per
step.  The issue peak and the butterfly ceiling are the round-1 microbenchmarks (profiles/r01_micro_valu_issue_rates.txt,
r01_micro_butterfly_rates.txt) at the kernel's occupancy of 4 waves/SIMD: v_mad_u64_u32-class instructions issue at 534 G wave-
instructions/s chip-wide, v_add_u32-class at 997 G; the hand-scheduled butterfly is 16 + 2 of them, and the register-resident
Shoup butterfly loop reaches 3.39 M (VGPR twiddles) .. 3.53 M (SGPR twiddles) limb-NTT(2^16)/s."""
import collections, csv, json, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from bench import csrc_tree_hash          # ties the counters to the kernel sources they were collected on (bench.py drops them when it differs)

agg = collections.defaultdict(lambda: collections.defaultdict(list))
for r in csv.DictReader(open(sys.argv[1])):
    agg[r["Kernel_Name"].split("(")[0].replace("void ", "")][r["Counter_Name"]].append(float(r["Counter_Value"]))
batch, launches = int(sys.argv[3]), int(sys.argv[4])
fused = [k for k in agg if "ntt_fwd_fused" in k]
assert len(fused) == 1, list(agg)
c = agg[fused[0]]
per_launch = sum(c["SQ_INSTS_VALU"]) / len(c["SQ_INSTS_VALU"])
slow, fast = 534.0, 997.0
mix_peak = 18.0 / (16.0 / slow + 2.0 / fast)
out = {
    "what": "VALU wave-instructions of one forward transform of the batch (%d launches of %s)" % (launches, fused[0]),
    "valu_wave_instructions_per_step": per_launch * launches,
    "valu_wave_instructions_per_poly": per_launch * launches / batch,
    "launches_sampled": len(c["SQ_INSTS_VALU"]),
    "issue_peak_Gwinstr_per_s": round(mix_peak, 1),
    "issue_peak_note": "16 v_mad_u64_u32-class (534 G/s) + 2 v_add_u32-class (997 G/s) per butterfly at 4 waves/SIMD, profiles/r01_micro_valu_issue_rates.txt",
    "butterfly_ceiling_limb_ntt_per_s": 3.46e6,
    "butterfly_ceiling_note": "register-resident Shoup butterfly loop at 4 workgroups/CU: 3.39 M (VGPR twiddles) .. 3.53 M (SGPR twiddles) limb-NTT(2^16)/s, half the stages each; profiles/r01_micro_butterfly_rates.txt",
    "other_counters_per_launch": {k: sum(v) / len(v) for k, v in c.items() if k != "SQ_INSTS_VALU"},
    "config": {"logn": 16, "limbs": 16, "batch": batch},
    "csrc_tree": csrc_tree_hash(),
    "source": "rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU -- python3 bench.py --no-cpu",
}
json.dump(out, open(sys.argv[2], "w"), indent=1)
print(json.dumps(out, indent=1))
