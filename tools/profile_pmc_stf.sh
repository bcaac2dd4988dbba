#!/bin/bash
# Runs on the GPU box: memory-path counters of k_linearize_stf (one counter group per rocprofv3 --pmc pass, no trace domains).
set -u
TAG=${1:-r02}
export HITL_STF_VARIANT=${2:-0}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmcstf_${TAG}
mkdir -p $OUT
i=0
for grp in "TA_TA_BUSY_sum TA_FLAT_READ_WAVEFRONTS_sum" "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" \
           "TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum" "TCP_TCC_READ_REQ_LATENCY_sum TCP_PENDING_STALL_CYCLES_sum" \
           "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_GATE_EN1_sum" \
           "TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_HIT_sum" \
           "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" "SQ_ACTIVE_INST_LDS SQ_WAVE_CYCLES" \
           "TCC_EA0_RDREQ_sum TCC_EA0_RDREQ_LEVEL_sum" "TCC_REQ_sum TCC_TAG_STALL_sum" \
           "SQ_INSTS_VALU SQ_WAVES" "SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY" \
           "GRBM_GUI_ACTIVE" "SQ_BUSY_CYCLES SQ_WAIT_ANY" \
           "SQ_INST_LEVEL_VMEM SQ_ACTIVE_INST_VMEM" "SQ_INSTS_VMEM_RD SQ_INST_LEVEL_LDS" \
           "TCC_HIT_sum TCC_MISS_sum" "TCC_EA0_RDREQ_32B_sum TCC_EA0_RDREQ_DRAM_sum"; do
  i=$((i+1))
  timeout 90 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/stf_time.py > $OUT/p$i.out 2> $OUT/p$i.err < /dev/null
done
python3 - <<PY
import csv, glob, os
from collections import defaultdict
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"].split("(")[0][:60]][row["Counter_Name"]].append(float(row["Counter_Value"]))
for k in sorted(vals):
    if "k_linearize_stf" in k or "k_assemble" in k or "k_stf_finalize" in k:
        print("##", k)
        for c in sorted(vals[k]):
            v = vals[k][c]
            print("  %-44s n=%3d mean=%.6g" % (c, len(v), sum(v) / len(v)))
PY
