#!/bin/bash
# round 4: parity of the default linearisation kernel + kernel-trace A/B of the variants given as arguments (HITL_STF_VARIANT)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-stfp}
shift
mkdir -p $OUT
timeout 1200 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_configs.py -m gpu -q > $OUT/pytest.txt 2>&1
tail -12 $OUT/pytest.txt
export TMPDIR=/tmp
for v in "$@"; do
  HITL_STF_VARIANT=$v timeout 120 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/v$v -- python3 tools/stf_time.py > $OUT/v$v.out 2> $OUT/v$v.err
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/v$v/*/*kernel_trace.csv"):
    d = {}
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        if "k_linearize_stf" in n or "k_stf_" in n: d.setdefault(n.split("(")[0][:60], []).append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for n, v in d.items():
        w = v[2:] if len(v) > 4 else v
        print("variant $v: %-60s n=%d avg %.1f us  min %.1f  max %.1f" % (n, len(w), sum(w) / len(w), min(w), max(w)))
PY
done 2>&1 | tee $OUT/ab.txt
