#!/bin/bash
# Runs on the GPU box (via gpurun): utilisation counters of bench.py's default workload, ONE counter group per pass
# (separate --pmc passes, no trace domains, as the pool requires).  Summarise with tools/summarize_pmc.py.
set -u
TAG=${1:-r01}
ARGS=${2:-"--steps 2 --warmup 1 --no-cpu-baseline"}
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OUT=gpurun_out/pmc_${TAG}
mkdir -p $OUT
i=0
for grp in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES" "SQ_WAIT_ANY SQ_WAIT_INST_ANY" "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY" "SQ_INSTS_VALU SQ_INSTS_SALU" \
           "SQ_INSTS_VMEM_RD SQ_INSTS_LDS" "GRBM_GUI_ACTIVE" "TCC_HIT_sum TCC_MISS_sum" "TCP_PENDING_STALL_CYCLES_sum TCP_GATE_EN1_sum" "SQ_WAVES"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $grp --output-format csv -d $OUT/p$i -- python3 bench.py $ARGS > $OUT/p$i.json 2> $OUT/p$i.err < /dev/null
done
python3 tools/summarize_pmc.py $OUT > $OUT/summary.md 2> $OUT/summary.err
cat $OUT/summary.md
