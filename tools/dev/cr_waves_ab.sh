#!/bin/bash
# Runs on the GPU box: rebuilds hitl_cr.o with -DHITL_CR_WAVES=<v> for each argument and prints the per-stage k_cr_chunk averages.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
for v in "$@"; do
  (cd hitl_slam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DHITL_CR_WAVES=$v -c hitl_cr.hip -o hitl_cr.o && make -s) || exit 1
  echo "== waves $v"
  tools/dev/cr_stage_ab.sh HITL_CR_THREADS - | tail -1
done
