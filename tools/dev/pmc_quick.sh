#!/bin/bash
# quick counter digest of the linearisation kernel under HITL_STF_VARIANT=$2 (a few rocprofv3 --pmc passes of tools/stf_time.py)
set -u
TAG=${1:-q}
export HITL_STF_VARIANT=${2:-0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmcq_${TAG}
mkdir -p $OUT
i=0
for grp in "TCC_HIT_sum TCC_MISS_sum" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" "SQ_BUSY_CYCLES SQ_WAIT_ANY" "SQ_INSTS_VALU SQ_WAVE_CYCLES" \
           "GRBM_GUI_ACTIVE" "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "TCC_EA0_RDREQ_sum TCP_TOTAL_CACHE_ACCESSES_sum" \
           "SQ_INST_LEVEL_VMEM SQ_INSTS_VMEM_RD" "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC" "SQ_WAIT_INST_LDS SQ_ACTIVE_INST_LDS" \
           "SQ_IFETCH SQ_IFETCH_LEVEL" "SQ_INSTS_LDS SQ_INSTS_VALU_MFMA_MOPS_F64" "SQ_ACTIVE_INST_ANY SQ_WAVES" "SQ_INST_CYCLES_VMEM SQ_INSTS_BRANCH" ${EXTRA_PMC:-}; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/stf_time.py > $OUT/p$i.out 2> $OUT/p$i.err < /dev/null
done
python3 - <<PY
import csv, glob
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    if "k_linearize_stf" in k:
        v = {c: sum(x[2:]) / max(len(x[2:]), 1) for c, x in vals[k].items()}
        print("##", k, "variant $HITL_STF_VARIANT")
        for c in sorted(v): print("  %-40s %.5g" % (c, v[c]))
        cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
        print("  -- cycles/launch %.0f; VALU busy %.1f %% of SIMD cycles; %.2f cyc/VALU instr; L2 hit %.1f %%; L1-miss latency %.0f cyc; wait_any %.1f %% wait_inst %.1f %% of wave cycles"
              % (cyc, 100 * 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / 1024 / max(cyc, 1), 4 * v.get("SQ_ACTIVE_INST_VALU", 0) / max(v.get("SQ_INSTS_VALU", 1), 1),
                 100 * v.get("TCC_HIT_sum", 0) / max(v.get("TCC_HIT_sum", 0) + v.get("TCC_MISS_sum", 0), 1),
                 v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(v.get("TCP_TCC_READ_REQ_sum", 1), 1),
                 100 * v.get("SQ_WAIT_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1), 100 * v.get("SQ_WAIT_INST_ANY", 0) / max(v.get("SQ_WAVE_CYCLES", 1), 1)))
PY
