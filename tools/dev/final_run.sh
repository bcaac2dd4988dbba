cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
mkdir -p gpurun_out/final_r04b
python bench.py > gpurun_out/final_r04b/bench_c4.json 2> gpurun_out/final_r04b/bench_c4.err
python bench.py --workload c2 > gpurun_out/final_r04b/bench_c2.json 2> gpurun_out/final_r04b/bench_c2.err
python bench.py --workload c5 --steps 3 --warmup 1 --no-cpu-baseline > gpurun_out/final_r04b/bench_c5.json 2> gpurun_out/final_r04b/bench_c5.err
bash tools/profile_bench.sh r04b > gpurun_out/final_r04b/profile.log 2>&1
tail -c 600 gpurun_out/final_r04b/bench_c4.json
echo
python - <<'PY'
import json
for w in ("c4","c2","c5"):
    try:
        r=json.loads(open("gpurun_out/final_r04b/bench_%s.json"%w).read().strip().splitlines()[-1])
        print(w, r["ms_per_step"], r["value"], r["lm_iterations_per_resolve"], r["roofline"]["frac"], r["roofline"]["avg_launch_ms"], r.get("interactive_correction_ms"))
    except Exception as e: print(w, "ERR", e)
PY
