"""Development helper: hitl_solver_find_correspondences with kd_tree_search on the C4 map; HITL_KD_BLOCK / HITL_KD_REFILL
select the workgroup size and refill threshold of k_match_kd_tgt (read once per process)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
opt = H.default_options(); opt.kd_tree_search = 1
s = H.Solver(opt)
s.set_map(off, px, py, nx, ny, cov)
ts = []
for rep in range(6):
    t0 = time.perf_counter()
    s.find_correspondences(init, 3)
    ts.append((time.perf_counter() - t0) * 1e3)
c = s.get_correspondences()
print("kd block=%s refill=%s: first %.2f ms, then min %.2f ms; pairs %d matches %d checksum %d" % (
    os.environ.get("HITL_KD_BLOCK", "-"), os.environ.get("HITL_KD_REFILL", "-"), ts[0], min(ts[1:]), len(c[0]), len(c[3]),
    int(np.asarray(c[3], np.int64).sum() * 31 + np.asarray(c[4], np.int64).sum())))
s.close()
