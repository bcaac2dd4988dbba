#!/bin/bash
# Development A/B of k_linearize_stf variants on the GPU box (HITL_STF_VARIANT = "src pipe minblocks", see launch_linearize_stf).
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-stfv}
mkdir -p $OUT
shift
for v in "$@"; do
  HITL_STF_VARIANT=$v python3 bench.py --steps 4 --warmup 1 --no-cpu-baseline > $OUT/v$v.json 2> $OUT/v$v.err
  python3 - <<PY
import json
try:
    r = json.load(open("$OUT/v$v.json"))
    print("variant $v: stf %.1f us/launch, resolve %.2f ms, iters %d, cost %.12g" % (r["roofline"]["avg_launch_ms"] * 1e3, r["ms_per_step"], r["lm_iterations_per_resolve"], r["final_cost"]))
except Exception as e:
    print("variant $v failed", e)
PY
done
