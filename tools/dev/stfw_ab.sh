#!/bin/bash
# round 4: parity of the wave-per-pair linearisation (HITL_STF_VARIANT >= 100) + kernel-trace A/B against the half-wave kernel
cd "${GRAFT_REPO_ROOT:-/root/repo}"
OUT=gpurun_out/${1:-stfw}
mkdir -p $OUT
shift
V=${1:-106}
HITL_STF_VARIANT=$V timeout 900 python3 -m pytest tests/test_gpu_parity.py tests/test_gpu_edge.py tests/test_gpu_configs.py -m gpu -x -q > $OUT/pytest_v$V.txt 2>&1
tail -5 $OUT/pytest_v$V.txt
bash tools/stf_trace.sh $(basename $OUT) "$@" 0 | tee $OUT/ab.txt
