"""Development helper: a few correspondence searches of the C4 workload (the reference's KD walk), for rocprofv3 runs on k_match_kd_tgt."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
s = H.Solver()
s.set_map(off, px, py, nx, ny, cov)
ts = []
for _ in range(int(os.environ.get("REPS", "4"))):
    t0 = time.perf_counter()
    s.find_correspondences(init, 3)
    ts.append((time.perf_counter() - t0) * 1e3)
print("search ms", " ".join("%.2f" % t for t in ts), "pairs, matches", s.correspondence_counts())
s.close()
