"""Development helper: 3 mode-C re-solves of a C4-like map with N poses (argv[1], default 50000) for kernel-trace runs."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n = int(sys.argv[1]) if len(sys.argv) > 1 else 50000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 400
mode = int(sys.argv[3]) if len(sys.argv) > 3 else 2
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, max(1, n // 1000), kind=0)
s = H.Solver()
s.set_map(off, px, py, nx, ny, cov)
s.set_human_constraints(rows)
if mode != 0:
    s.find_correspondences(init, 3)
for _ in range(3):
    out, sm = s.solve(mode, init)
print("n", n, "iterations", sm.iterations, "total_ms", sm.total_ms)
