#!/bin/bash
# round-end artefacts in one call: GPU tests, the three bench lines, kernel trace + traffic counters, utilisation counters
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r05}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gputests.txt 2>&1
tail -3 $OUT/gputests.txt
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1
bash tools/profile_pmc.sh $TAG > $OUT/pmc.log 2>&1
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --workload c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python - <<PY
import json
for w in ("c4","c2","c5"):
    try:
        r=json.loads(open("$OUT/bench_%s.json"%w).read().strip().splitlines()[-1])
        print(w, "ms/step %.3f value %.4g iters %d frac %.3f stf %.1f us fp64 %s interactive %s" % (r["ms_per_step"], r["value"], r["lm_iterations_per_resolve"], r["roofline"]["frac"], 1e3*r["roofline"]["avg_launch_ms"], r["roofline"].get("fp64_issue_frac"), r.get("interactive_correction_ms")))
    except Exception as e: print(w, "ERR", e)
PY
