#!/bin/bash
# Memory-path counters of k_linearize_stf_r (why is the texture-address unit 70 % busy?), one group per rocprofv3 --pmc pass.
# usage: pmc_stf_mem.sh <tag> [ORDER]   (ORDER = src | tgt | rand: tools/dev/stf_order_ab.py)
set -u
TAG=$1; export ORDER=${2:-src}; export REPS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmcstf_${TAG}
mkdir -p $OUT
i=0
for grp in "TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "TCP_TCC_READ_REQ_sum TCP_TOTAL_READ_sum" "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" \
           "TA_ADDR_STALLED_BY_TC_CYCLES_sum TA_DATA_STALLED_BY_TC_CYCLES_sum" "TCP_GATE_EN1_sum TCP_GATE_EN2_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_TA_TCP_STATE_READ_sum" \
           "TCP_TCC_READ_REQ_LATENCY_sum TCP_TCP_LATENCY_sum" "TCP_READ_TAGCONFLICT_STALL_CYCLES_sum TCP_TCR_TCP_STALL_CYCLES_sum" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" \
           "TA_TOTAL_WAVEFRONTS_sum TA_FLAT_WAVEFRONTS_sum" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" "TCP_TCP_TA_DATA_STALL_CYCLES_sum TCP_TD_TCP_STALL_CYCLES_sum" "TCC_HIT_sum TCC_MISS_sum" "SQ_WAVES SQ_WAVE_CYCLES"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 tools/dev/stf_order_ab.py > $OUT/p$i.out 2> $OUT/p$i.err < /dev/null
done
python3 - <<PY
import csv, glob
from collections import defaultdict
vals = defaultdict(list)
for f in glob.glob("$OUT/p*/*/*counter_collection.csv"):
    for row in csv.DictReader(open(f)):
        if "k_linearize_stf" in row["Kernel_Name"]: vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
v = {c: sum(x[1:]) / max(len(x[1:]), 1) for c, x in vals.items()}
print("## k_linearize_stf_r, match order $ORDER (per launch, averages over %d launches)" % max(len(x) - 1 for x in vals.values()))
for c in sorted(v): print("  %-44s %.6g" % (c, v[c]))
g = v.get("TA_FLAT_READ_WAVEFRONTS_sum", 0); cyc = v.get("GRBM_GUI_ACTIVE", 0) / 8
if g:
    print("  -- vector read instructions (wave level) %.4g; tag look-ups per read instruction %.2f; L1->L2 read requests per read instruction %.2f; L1 hit rate %.1f %%"
          % (g, v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 0) / g, v.get("TCP_TCC_READ_REQ_sum", 0) / g, 100 * (1 - v.get("TCP_TCC_READ_REQ_sum", 0) / max(v.get("TCP_TOTAL_CACHE_ACCESSES_sum", 1), 1))))
    print("  -- kernel cycles %.0f; TA busy %.1f %% (256 TAs); TA stalled by TC: address %.1f %%, data %.1f %%; TCP busy (GATE_EN2 / GATE_EN1) %.1f %%; mean L1->L2 read latency %.0f cycles"
          % (cyc, 100 * v.get("TA_TA_BUSY_sum", 0) / 256 / max(cyc, 1), 100 * v.get("TA_ADDR_STALLED_BY_TC_CYCLES_sum", 0) / 256 / max(cyc, 1), 100 * v.get("TA_DATA_STALLED_BY_TC_CYCLES_sum", 0) / 256 / max(cyc, 1),
             100 * v.get("TCP_GATE_EN2_sum", 0) / max(v.get("TCP_GATE_EN1_sum", 1), 1), v.get("TCP_TCC_READ_REQ_LATENCY_sum", 0) / max(v.get("TCP_TCC_READ_REQ_sum", 1), 1)))
PY
