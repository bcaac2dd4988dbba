#!/bin/bash
# VERDICT r5 item 3 (counters route): vector-issue, wait and texture-address counters of k_linearize_stf_r as shipped (STF_LDS=0) and with its
# targets read from LDS (STF_LDS=1, tools/dev/stf_lds_targets_ab.py), one counter group per rocprofv3 --pmc pass, same box, same process layout.
# usage: pmc_stf_ab.sh <tag>
set -u
TAG=$1; export REPS=1
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
export TMPDIR=/tmp
OUT=gpurun_out/pmcstfab_${TAG}
mkdir -p $OUT
python3 tools/dev/stf_lds_targets_ab.py > $OUT/times.txt 2> $OUT/times.err
for v in 0 1; do
  export STF_LDS=$v
  i=0
  for grp in "TA_TA_BUSY_sum GRBM_GUI_ACTIVE" "SQ_WAIT_ANY SQ_WAVE_CYCLES" "SQ_ACTIVE_INST_VALU SQ_BUSY_CYCLES" "SQ_INSTS_VALU SQ_INSTS_VMEM_RD" "SQ_INSTS_LDS SQ_ACTIVE_INST_LDS" \
             "SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY" "SQ_INST_CYCLES_VMEM SQ_WAVES" "TA_FLAT_READ_WAVEFRONTS_sum TCP_TOTAL_CACHE_ACCESSES_sum" "SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
             "SQ_INSTS_SALU SQ_INSTS_SMEM" "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_MISC"; do
    i=$((i+1))
    timeout 200 rocprofv3 --pmc $grp --output-format csv -d $OUT/v${v}_p$i -- python3 tools/dev/stf_lds_targets_ab.py > $OUT/v${v}_p$i.out 2> $OUT/v${v}_p$i.err < /dev/null
  done
done
python3 - <<PY > $OUT/summary.md
import csv, glob
from collections import defaultdict
print(open("$OUT/times.txt").read())
tab = {}
for v in (0, 1):
    vals = defaultdict(list)
    for f in glob.glob("$OUT/v%d_p*/*/*counter_collection.csv" % v):
        for row in csv.DictReader(open(f)):
            if "k_linearize_stf_r" in row["Kernel_Name"]: vals[row["Counter_Name"]].append(float(row["Counter_Value"]))
    tab[v] = {c: sum(x[1:]) / max(len(x[1:]), 1) for c, x in vals.items()}
names = sorted(set(tab[0]) | set(tab[1]))
print("| counter (per launch) | shipped (L1 gathers) | targets from LDS | ratio |")
print("|---|---|---|---|")
for c in names:
    a, b = tab[0].get(c, float("nan")), tab[1].get(c, float("nan"))
    print("| %s | %.5g | %.5g | %.3f |" % (c, a, b, b / a if a else float("nan")))
for v in (0, 1):
    t = tab[v]; cyc = t.get("GRBM_GUI_ACTIVE", 0) / 8
    print("\nvariant %d: kernel cycles %.0f; TA busy %.1f %% of 256 units; SQ_WAIT_ANY / SQ_WAVE_CYCLES %.1f %%; SQ_ACTIVE_INST_VALU / SQ_BUSY_CYCLES %.3f; SQ_WAIT_INST_ANY / SQ_WAVE_CYCLES %.1f %%"
          % (v, cyc, 100 * t.get("TA_TA_BUSY_sum", 0) / 256 / max(cyc, 1), 100 * t.get("SQ_WAIT_ANY", 0) / max(t.get("SQ_WAVE_CYCLES", 1), 1),
             t.get("SQ_ACTIVE_INST_VALU", 0) / max(t.get("SQ_BUSY_CYCLES", 1), 1), 100 * t.get("SQ_WAIT_INST_ANY", 0) / max(t.get("SQ_WAVE_CYCLES", 1), 1)))
PY
cat $OUT/summary.md
