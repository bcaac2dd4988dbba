"""Development helper: 12 linearisations of the C4 workload at a fixed state (hitl_solver_linearize), for rocprofv3 --kernel-trace --stats
A/B runs of k_linearize_stf variants (HITL_STF_VARIANT); results of ablation variants are deliberately wrong and ignored."""
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
x = init.astype(np.float64)
for _ in range(12):
    s.linearize(2, init, x)
