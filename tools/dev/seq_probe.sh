#!/bin/bash
# Runs on the GPU box: development build of hitl_cr.o with probes (-DHITL_CR_PROBE), then the per-workgroup timeline of k_cr_seq on C4:
# when every workgroup started and ended (100 MHz clock), on which XCD / CU.  The shipped library is restored afterwards.
cd "${GRAFT_REPO_ROOT:-/root/repo}" || exit 1
(cd hitl_slam_amd/csrc && /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-result -DHITL_CR_PROBE $EXTRA -c hitl_cr.hip -o hitl_cr.o && make -s) || exit 1
SEQ_PROBE=1 REPS=2 python3 tools/dev/solve_c4.py 2>&1 | tail -40
(cd hitl_slam_amd/csrc && rm -f hitl_cr.o && make -s)
