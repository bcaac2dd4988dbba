#!/bin/bash
# copies the round-end artefacts of tools/dev/final_all.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names
# usage: collect_profiles.sh r05
TAG=${1:-r05}
cd "$(dirname "$0")/../.." || exit 1
P=gpurun_out/prof_$TAG; C=gpurun_out/pmc_$TAG; F=gpurun_out/final_$TAG
cp $P/summary.md profiles/${TAG}_bench_c4_summary.md
cp "$(ls $P/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_c4_kernel_stats.csv
cp $C/summary.md profiles/${TAG}_pmc_c4.md
python3 - <<PY
import json
t = json.load(open("$P/traffic.json"))
t["note"] = "round ${TAG#r}: k_linearize_stf_r<7>: FETCH_SIZE %.1f KiB x2 + WRITE_SIZE %.1f KiB per launch; kernel-trace average %.1f us, HIP-event average %.1f us in the profiled run" % (
    t["fetch_kib"], t["write_kib"], t["stf_avg_us"], 1e3 * t["bench_avg_launch_ms"])
json.dump(t, open("profiles/traffic_c4.json", "w"), indent=1)
for w in ("c4", "c2", "c5"):
    line = [l for l in open("$F/bench_%s.json" % w).read().strip().splitlines() if l.startswith("{")][-1]
    open("profiles/${TAG}_final_bench_%s.json" % (w if w != "c5" else "c5_1gpu"), "w").write(json.dumps(json.loads(line), indent=1) + "\n")
PY
tail -4 $F/gputests.txt > profiles/${TAG}_final_gputests.txt
ls -la profiles/${TAG}_*
