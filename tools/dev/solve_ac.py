"""Development helper: C4 re-solves in mode A and mode C (3 + 3), for kernel-trace A/B runs of two builds of the library (HITL_LIBHITLGPU)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
from hitl_slam_amd.api import apply_test_switches_from_env
apply_test_switches_from_env()          # HITL_TEST_SWITCHES="cr_seq=0": the library's test hooks, set by this script
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
s = H.Solver()
s.set_map(off, px, py, nx, ny, cov)
s.set_human_constraints(S.constraints(p, gt, 50, kind=4))
s.find_correspondences(init, 3)
for mode in (H.MODE_ODOM_HC, H.MODE_JOINT):
    ts = []
    for _ in range(4):
        t0 = time.perf_counter()
        poses, summ = s.solve(mode, init)
        ts.append((time.perf_counter() - t0) * 1e3)
    print("mode", mode, "solve ms", " ".join("%.2f" % t for t in ts), "iterations", summ.iterations)
s.close()
