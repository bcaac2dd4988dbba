#!/bin/bash
# round-end artefacts in one call: GPU tests, the bench lines (c4 = headline, c2, c5, c3 = the PCG workload), kernel trace + traffic counters
# of c4 and c3, utilisation counters of c4, the 4- and 8-submap splits with ranks sharing the one GPU over gloo
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
TAG=${1:-r06}
OUT=gpurun_out/final_$TAG
mkdir -p $OUT
timeout 2400 python -m pytest tests -q -m gpu > $OUT/gputests.txt 2>&1
tail -3 $OUT/gputests.txt
bash tools/profile_bench.sh $TAG > $OUT/profile.log 2>&1
bash tools/profile_pmc.sh $TAG > $OUT/pmc.log 2>&1
bash tools/profile_bench.sh ${TAG}c3 "--workload c3 --steps 3 --warmup 1 --no-cpu-baseline" > $OUT/profile_c3.log 2>&1
python3 tools/summarize_pcg_profile.py gpurun_out/prof_${TAG}c3 > $OUT/traffic_c3.log 2>&1
cp gpurun_out/prof_${TAG}c3/traffic_c3.json profiles/traffic_c3.json      # (on the box: so that the c3 line below carries `traffic`; collect_profiles.sh copies it here)
python bench.py > $OUT/bench_c4.json 2> $OUT/bench_c4.err
python bench.py --workload c2 > $OUT/bench_c2.json 2> $OUT/bench_c2.err
python bench.py --workload c5 > $OUT/bench_c5.json 2> $OUT/bench_c5.err
python bench.py --workload c3 > $OUT/bench_c3.json 2> $OUT/bench_c3.err
for spec in "4 c4" "8 c5"; do
  set -- $spec
  timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node $1 --master-addr 127.0.0.1 --master-port 29533 bench.py --gpus $1 --workload $2 --backend gloo \
      --steps 3 --warmup 1 --no-cpu-baseline > $OUT/bench_$2_$1submaps.json 2> $OUT/bench_$2_$1submaps.err
done
python - <<PY
import json
for w in ("c4","c2","c5","c3","c4_4submaps","c5_8submaps"):
    try:
        r=json.loads([l for l in open("$OUT/bench_%s.json"%w).read().strip().splitlines() if l.startswith("{")][-1])
        print(w, "ms/step %.3f value %.4g iters %d frac %.4f avg launch %.1f us traffic %s interactive %s search %s" % (r["ms_per_step"], r["value"], r["lm_iterations_per_resolve"], r["roofline"]["frac"],
              1e3*r["roofline"]["avg_launch_ms"], r["roofline"].get("traffic"), r.get("interactive_correction_ms"), r.get("correspondence_search_ms")))
    except Exception as e: print(w, "ERR", e)
PY
