import os, sys, time
import numpy as np
sys.path.insert(0, "/root/repo")
import hitl_slam_amd as H
from tools import hitl_synth as S
p = S.params(1000, 360, seed=20262102, poses_per_floor=200)
gt, init, cov = S.poses(p); off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, 2, kind=4)
s = H.Solver(); s.set_map(off, px, py, nx, ny, cov); s.set_human_constraints(rows)
t0 = time.time(); s.find_correspondences(init, 0); t1 = time.time() - t0
P, C = s.correspondence_counts()
print("all-pairs search on 1k x 360: %.1f ms, %d pairs, %d matches" % (t1 * 1e3, P, C))
t0 = time.time(); out, summ = s.solve(H.MODE_JOINT, init); t2 = time.time() - t0
print("mode C solve: %.1f ms, iters %d, term %d, offband %d, linear iters %d, cost %.6g -> %.6g" % (t2 * 1e3, summ.iterations, summ.termination, summ.num_offband_pairs, summ.linear_iterations, summ.initial_cost, summ.final_cost))
