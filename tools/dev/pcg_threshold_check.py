import sys, numpy as np
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import hitl_slam_amd as H
from oracle import hitl_oracle as O
from conftest import Problem
import os
for n in (10900, 11000, 2000):
    prob = Problem(n, 48, ppf=200, groups=max(1, n // 1000), kind=0)
    opt = H.default_options(); opt.max_band_poses = 1; opt.max_linear_iterations = int(os.environ.get('MAXLIN', 50))
    g, o = H.Solver(opt), O.OracleSolver(opt)
    prob.load(g); prob.load(o)
    o.find_correspondences(prob.init, 3, 0)
    g.set_correspondences(*o.get_correspondences())
    pg, sg = g.solve(2, prob.init); po, so = o.solve(2, prob.init)
    print(n, "gpu", sg.iterations, sg.termination, "%.9g" % sg.final_cost, "lin", sg.linear_iterations, "offband", sg.num_offband_pairs, "| oracle", so.iterations, so.termination, "%.9g" % so.final_cost,
          "| dxy %.2e dth %.2e" % (np.abs(pg[:, :2] - po[:, :2]).max(), np.abs(pg[:, 2] - po[:, 2]).max()))
    g.close(); o.close()
