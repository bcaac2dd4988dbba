#!/bin/bash
# GPU box: the full (not the quick) size sweep of tools/cr_size_sweep.py through the fused stages, the level-per-launch
# kernels and the submap code path on one rank; prints every size on which the three disagree.  Also band widths 2 and 4.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
mkdir -p gpurun_out/sweep
for w in "" 2 4; do
  for tag in fused legacy dist; do
    unset HITL_CR_LEGACY HITL_FORCE_DIST HITL_SWEEP_WINDOW
    [ "$tag" = legacy ] && export HITL_CR_LEGACY=1
    [ "$tag" = dist ] && export HITL_FORCE_DIST=1
    [ -n "$w" ] && export HITL_SWEEP_WINDOW=$w
    timeout 1500 python3 tools/cr_size_sweep.py gpurun_out/sweep/${tag}_w${w:-3}.jsonl > /dev/null 2>&1
  done
  python3 - <<PY
import json
w = "${w:-3}"
r = {t: [json.loads(l) for l in open("gpurun_out/sweep/%s_w%s.jsonl" % (t, w))] for t in ("fused", "legacy", "dist")}
bad = 0
for a, b, c in zip(r["fused"], r["legacy"], r["dist"]):
    ok = "err" not in a and "err" not in b and "err" not in c and a["it"] == b["it"] == c["it"] and a["term"] == b["term"] == c["term"] and abs(a["cost"] - b["cost"]) <= 1e-9 * max(abs(b["cost"]), 1e-12)
    if not ok: bad += 1; print("MISMATCH", a, b, c)
print("window %s: %d sizes, %d mismatches" % (w, len(r["fused"]), bad))
PY
done
