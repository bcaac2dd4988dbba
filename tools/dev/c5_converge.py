"""Development helper (GPU box): how many LM iterations the C5 map needs to reach CONVERGENCE."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import numpy as np
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 200000, 512
p = S.params(n, m, seed=20265102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, 200, kind=0)
opt = H.default_options()
opt.max_num_iterations = 2000
g = H.Solver(opt)
g.set_map(off, px, py, nx, ny, cov)
g.set_human_constraints(rows)
t0 = time.time(); g.find_correspondences(init, 3); t1 = time.time(); g.find_correspondences(init, 3); t2 = time.time()
print("search %.1f ms first, %.1f ms again; counts %s" % ((t1 - t0) * 1e3, (t2 - t1) * 1e3, g.correspondence_counts()))
t0 = time.time()
pg, sg = g.solve(H.MODE_JOINT, init)
print("C5: term %d iters %d ok %d cost %.12g -> %.12g in %.1f ms" % (sg.termination, sg.iterations, sg.successful_steps, sg.initial_cost, sg.final_cost, (time.time() - t0) * 1e3))
