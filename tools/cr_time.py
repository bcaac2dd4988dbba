"""Development helper: 8 factorisations of the C4 system (hitl_solver_pose_covariances -> launch_cr_factor + selected inversion) for
rocprofv3 --kernel-trace A/B runs of the fused cyclic reduction (HITL_CR_ABL ablations give wrong numbers on purpose)."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
s = H.Solver()
s.set_map(off, px, py, nx, ny, cov)
s.find_correspondences(init, 3)
for _ in range(8):
    try:
        s.pose_covariances(H.MODE_JOINT, init)
    except Exception as e:
        pass
