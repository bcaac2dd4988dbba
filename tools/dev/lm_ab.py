"""Development helper (GPU box): device-driven vs host-driven LM loop on the C4 bench workload, per-call wall clock."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, 50, kind=0)
for host in (0, 1, 0, 1):
    if not host: os.environ["HITL_DEVICE_LM"] = "1"
    else: os.environ.pop("HITL_DEVICE_LM", None)
    s = H.Solver()
    s.set_map(off, px, py, nx, ny, cov)
    s.set_human_constraints(rows)
    s.find_correspondences(init, 3)
    for _ in range(2): s.solve(H.MODE_JOINT, init)
    t_hc, t_solve, t_tot = [], [], []
    for _ in range(8):
        t0 = time.perf_counter(); s.set_human_constraints(rows); t1 = time.perf_counter()
        out, sm = s.solve(H.MODE_JOINT, init); t2 = time.perf_counter()
        t_hc.append((t1 - t0) * 1e3); t_solve.append((t2 - t1) * 1e3); t_tot.append(sm.total_ms)
    print("host_lm=%d: set_hc %.3f ms, solve %.3f ms (min %.3f), summary.total_ms %.3f, iters %d" % (host, np.median(t_hc), np.median(t_solve), min(t_solve), np.median(t_tot), sm.iterations))
    s.close()
