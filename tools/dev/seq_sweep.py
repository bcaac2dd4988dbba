"""Development helper: linear-solve time per LM iteration (HIP events, timing_level 2) of C4 in modes A and C for several chunk sizes of
the two-way sequential first stage (test hook cr_seq_m) and for the cyclic-reduction first stage (cr_seq = 0)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = int(os.environ.get("POSES", 50000)), 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, max(1, n // 1000), kind=4)
corr = None
for tag, hooks in [("cr", dict(cr_seq=0))] + [("seq%d" % k, dict(cr_seq=1, cr_seq_m=k)) for k in (8, 12, 16, 17, 18, 20, 24, 0)]:
    H.set_test_switch(None)
    for k, v in hooks.items():
        H.set_test_switch(k, v)
    o = H.default_options(); o.timing_level = 2
    s = H.Solver(o)
    s.set_map(off, px, py, nx, ny, cov)
    s.set_human_constraints(rows)
    if corr is None:
        s.find_correspondences(init, 3); corr = s.get_correspondences()
    else:
        s.set_correspondences(*corr)
    out = []
    for mode in (H.MODE_ODOM_HC, H.MODE_JOINT):
        s.solve(mode, init)
        ts = []
        for _ in range(3):
            t0 = time.perf_counter(); _, sm = s.solve(mode, init); ts.append((time.perf_counter() - t0) * 1e3)
        out.append("mode %d: %.1f us/iter linear solve, solve %.2f ms, %d it, cost %.12g" % (mode, 1e3 * sm.linear_solver_ms / max(sm.iterations, 1), min(ts), sm.iterations, sm.final_cost))
    print("%-6s %s" % (tag, " | ".join(out)), flush=True)
    s.close()
