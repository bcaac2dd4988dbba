#!/bin/bash
# Runs on the GPU box: development build of hitl_cr.o with phase probes (-DHITL_CR_PROBE), then the shader-clock differences
# between the phase boundaries of one workgroup per fused stage (k_cr_chunk) on the LM path of C4.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
(cd hitl_slam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DHITL_CR_PROBE $EXTRA -c hitl_cr.hip -o hitl_cr.o && make -s) || exit 1
CR_PROBE=1 REPS=2 python3 tools/dev/solve_c4.py 2>&1 | tail -5
