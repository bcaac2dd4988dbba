"""Development helper: the C2 corridor (1000 poses x 360 observations) search + re-solves, for kernel traces of the small-map path."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 1000, 360
p = S.params(n, m, seed=20264102, poses_per_floor=1000)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
s = H.Solver()
s.set_map(off, px, py, nx, ny, cov)
s.set_human_constraints(S.constraints(p, gt, 5, kind=4))
s.find_correspondences(init, 3)
ts = []
for _ in range(6):
    t0 = time.perf_counter(); poses, summ = s.solve(H.MODE_JOINT, init); ts.append((time.perf_counter() - t0) * 1e3)
print("solve ms", " ".join("%.3f" % t for t in ts), "iterations", summ.iterations)
s.close()
