#!/bin/bash
# Runs on the GPU box (via gpurun): kernel-trace stats + separate PMC passes for HBM traffic of
# bench.py's default workload.  Outputs under gpurun_out/prof_<tag>/ ; summarise with
# tools/summarize_profile.py and copy the summaries into profiles/.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 3 --warmup 1 --no-cpu-baseline"}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/prof_${TAG}
mkdir -p $OUT
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/trace -- python3 bench.py $ARGS > $OUT/bench_trace.json 2> $OUT/trace.err < /dev/null
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $OUT/pmc_fetch -- python3 bench.py $ARGS > $OUT/bench_pmc_fetch.json 2> $OUT/pmc_fetch.err < /dev/null
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $OUT/pmc_write -- python3 bench.py $ARGS > $OUT/bench_pmc_write.json 2> $OUT/pmc_write.err < /dev/null
python3 tools/summarize_profile.py $OUT > $OUT/summary.md 2> $OUT/summary.err
ls -R $OUT | head -50
