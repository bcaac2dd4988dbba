"""Development helper: one mode-C re-solve of the c3 workload (Figure-8 stand-in, all pairs) with HITL_TIMING=1: the persistent PCG launch
prints where one CG iteration's time goes, per workgroup role (k_pcg_fused probe).  usage: pcg_probe.py [poses] [obs] [laps]"""
import os, sys
os.environ["HITL_TIMING"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hitl_slam_amd as H
from hitl_slam_amd.api import hc_rows
from tools import figure8
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 360
laps = int(sys.argv[3]) if len(sys.argv) > 3 else 4
d = figure8.make(n, m, seed=7, laps=laps)
opt = H.default_options(); opt.timing_level = 2
s = H.Solver(opt)
s.set_map(d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
s.set_human_constraints(hc_rows(figure8.constraints(d["gt"], 5, 3)))
s.find_correspondences(d["init"], 0)
for k in range(2):
    p, sm = s.solve(H.MODE_JOINT, d["init"])
cg = sm.linear_iterations - sm.iterations
print("re-solve: %d LM, %d CG iterations, %.2f us per CG iteration, linear solver %.3f ms per LM iteration, total %.2f ms" % (
    sm.iterations, cg, 1e3 * sm.pcg_ms / max(sm.pcg_timed_iterations, 1), sm.linear_solver_ms / sm.iterations, sm.total_ms))
ps, pt = s.get_correspondences()[:2]
far = np.abs(ps.astype(np.int64) - pt) > sm.band_poses
deg = np.zeros(n, np.int64)
und = set()
for a_, b_ in zip(ps[far], pt[far]):
    und.add((min(a_, b_), max(a_, b_)))
for a_, b_ in und:
    deg[a_] += 1; deg[b_] += 1
print("off-band partners per pose: mean %.1f, max %d (pose %d), first 16 poses: %s; poses with more than 55: %d" % (
    deg.mean(), deg.max(), int(deg.argmax()), deg[:16].tolist(), int((deg > 55).sum())))
