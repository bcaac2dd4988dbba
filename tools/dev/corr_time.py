"""Development helper: timing breakdown of hitl_solver_find_correspondences on the C4 map (HITL_TIMING=1 prints the phases)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
for kd in (0, 1):
    opt = H.default_options(); opt.kd_tree_search = kd
    s = H.Solver(opt)
    s.set_map(off, px, py, nx, ny, cov)
    for rep in range(8):
        t0 = time.perf_counter()
        s.find_correspondences(init, 3)
        print("kd=%d rep %d: %.2f ms" % (kd, rep, (time.perf_counter() - t0) * 1e3), file=sys.stderr)
    s.close()
