#!/bin/bash
# copies the round-end artefacts of tools/dev/final_all.sh <tag> from gpurun_out/ (scratch) into profiles/ (tracked) under the round's names
# usage: collect_profiles.sh r06
TAG=${1:-r06}
cd "$(dirname "$0")/../.." || exit 1
P=gpurun_out/prof_$TAG; P3=gpurun_out/prof_${TAG}c3; C=gpurun_out/pmc_$TAG; F=gpurun_out/final_$TAG
cp $P/summary.md profiles/${TAG}_bench_c4_summary.md
cp "$(ls -t $P/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_c4_kernel_stats.csv
cp $C/summary.md profiles/${TAG}_pmc_c4.md
if [ -d $P3 ]; then
  cp $P3/summary.md profiles/${TAG}_bench_c3_summary.md
  cp "$(ls -t $P3/trace/*/*kernel_stats.csv | head -1)" profiles/${TAG}_bench_c3_kernel_stats.csv
  cp $P3/traffic_c3.json profiles/traffic_c3.json
  cat $F/traffic_c3.log >> profiles/${TAG}_bench_c3_summary.md
fi
python3 - <<PY
import json, os
t = json.load(open("$P/traffic.json"))
t["note"] = "round ${TAG#r}: k_linearize_stf_r<7>: FETCH_SIZE %.1f KiB x2 + WRITE_SIZE %.1f KiB per launch; kernel-trace average %.1f us, HIP-event average %.1f us in the profiled run" % (
    t["fetch_kib"], t["write_kib"], t["stf_avg_us"], 1e3 * t["bench_avg_launch_ms"])
json.dump(t, open("profiles/traffic_c4.json", "w"), indent=1)
names = {"c4": "final_bench_c4", "c2": "final_bench_c2", "c5": "final_bench_c5_1gpu", "c3": "final_bench_c3",
         "c4_4submaps": "bench_c4_4submaps_shared_gpu", "c5_8submaps": "bench_c5_8submaps_shared_gpu"}
for w, nm in names.items():
    f = "$F/bench_%s.json" % w
    if not os.path.exists(f): continue
    lines = [l for l in open(f).read().strip().splitlines() if l.startswith("{")]
    if not lines: continue
    open("profiles/${TAG}_%s.json" % nm, "w").write(json.dumps(json.loads(lines[-1]), indent=1) + "\n")
PY
tail -4 $F/gputests.txt > profiles/${TAG}_final_gputests.txt
ls -la profiles/${TAG}_* profiles/traffic_c*.json
