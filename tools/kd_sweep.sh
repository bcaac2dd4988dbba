#!/bin/bash
# GPU box: parity of the KD search first, then the workgroup-size / refill sweep of k_match_kd_tgt on C4
mkdir -p gpurun_out
python -m pytest tests/test_gpu_parity.py -m gpu -x -q -k "kd" 2>&1 | tail -5
for blk in ${BLOCKS:-128 256}; do for rf in ${REFILLS:-8 16 24}; do
  HITL_KD_BLOCK=$blk HITL_KD_REFILL=$rf python tools/kd_time.py 2>&1 | grep "^kd block"
done; done | tee gpurun_out/kd_sweep.txt
HITL_TIMING=1 HITL_KD_BLOCK=256 HITL_KD_REFILL=16 python tools/kd_time.py 2>&1 | tail -4
