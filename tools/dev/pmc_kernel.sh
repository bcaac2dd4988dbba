#!/bin/bash
# counter digest of ONE kernel: usage pmc_kernel.sh <tag> <kernel name substring> <python script> (one counter group per rocprofv3 --pmc pass)
set -u
TAG=$1; KERN=$2; SCRIPT=$3
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmck_${TAG}
mkdir -p $OUT
i=0
for grp in "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_WAVE_CYCLES" "GRBM_GUI_ACTIVE" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" "SQ_INSTS_LDS SQ_INSTS_SALU" "SQ_ACTIVE_INST_ANY SQ_WAVES" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR" "SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH" "SQ_THREAD_CYCLES_VALU SQ_INST_CYCLES_SALU" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 120 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 $SCRIPT > $OUT/p$i.out 2> $OUT/p$i.err < /dev/null
done
python3 - <<PY
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    if "$KERN" in k:
        v = {c: sum(x[1:]) / max(len(x[1:]), 1) for c, x in vals[k].items()}
        print("##", k)
        for c in sorted(v): print("  %-40s %.5g" % (c, v[c]))
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
        print("  -- cycles/launch %.0f; VALU busy %.1f %% of SIMD cycles; %.2f cyc/VALU instr; lanes active per VALU instr %.1f; LDS busy %.1f %%; bank-conflict share of LDS cycles %.1f %%; wait_any %.1f %% wait_inst %.1f %% of wave cycles"
              % (cyc, 100 * 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / max(cyc, 1), 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_INSTS_VALU", 1), 1),
                 v.get("SQ_THREAD_CYCLES_VALU", 0) / max(4 * v.get("SQ_ACTIVE_INST_VALU", 1), 1),
                 100 * 4 * v.get("SQ_ACTIVE_INST_LDS", 0) / 1024 / max(cyc, 1), 100 * v.get("SQ_LDS_BANK_CONFLICT", 0) / max(v.get("SQ_LDS_IDX_ACTIVE", 1), 1),
                 100 * v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 100 * v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
PY
