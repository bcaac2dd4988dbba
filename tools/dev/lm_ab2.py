"""Development helper (GPU box): device- vs host-driven LM loop, several workloads (median solve wall clock)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hitl_slam_amd as H
from tools import hitl_synth as S
for name, n, m, groups, kind, moments in (("c2", 1000, 360, 2, 4, 0), ("c4 moments", 50000, 400, 50, 0, 1), ("10k x 200", 10000, 200, 10, 0, 0), ("c4 mode A", 50000, 400, 50, 0, 2)):
    p = S.params(n, m, seed=20262102, poses_per_floor=200)
    gt, init, cov = S.poses(p)
    off, px, py, nx, ny = S.scans(p)
    rows = S.constraints(p, gt, groups, kind=kind)
    for host in (0, 1):
        if not host: os.environ["HITL_DEVICE_LM"] = "1"
        else: os.environ.pop("HITL_DEVICE_LM", None)
        o = H.default_options(); o.stf_moments = 1 if moments == 1 else 0
        s = H.Solver(o)
        s.set_map(off, px, py, nx, ny, cov); s.set_human_constraints(rows)
        mode = H.MODE_ODOM_HC if moments == 2 else H.MODE_JOINT
        if mode == H.MODE_JOINT: s.find_correspondences(init, 3)
        for _ in range(3): s.solve(mode, init)
        ts = []
        for _ in range(15):
            t0 = time.perf_counter(); out, sm = s.solve(mode, init); ts.append((time.perf_counter() - t0) * 1e3)
        print("%-12s host_lm=%d: solve %.3f ms (min %.3f), iters %d -> %.1f us/iter" % (name, host, np.median(ts), min(ts), sm.iterations, 1e3 * np.median(ts) / max(sm.iterations, 1)))
        s.close()
