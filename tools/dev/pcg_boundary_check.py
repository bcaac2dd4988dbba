"""Development helper: the PCG preconditioner solve (k_cr_chunk_fwd on stored factors) at a chunk-boundary size of the fused
cyclic reduction (nsb = 1034: the last super-block is local node M of the last second-stage chunk) against the oracle."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import hitl_slam_amd as H
from oracle import hitl_oracle as O
from conftest import Problem
for n, band in ((3100, 3), (3120, 3), (1030, 1), (2060, 2)):
    prob = Problem(n, 48, ppf=200, groups=3, kind=0)
    opt = H.default_options(); opt.max_band_poses = band
    g, o = H.Solver(opt), O.OracleSolver(opt)
    prob.load(g); prob.load(o)
    o.find_correspondences(prob.init, band + 2, 0)
    g.set_correspondences(*o.get_correspondences())
    pg, sg = g.solve(2, prob.init, allow_failure=True); po, so = o.solve(2, prob.init)
    print(n, band, "gpu", sg.iterations, sg.termination, "%.9g" % sg.final_cost, "CG", sg.linear_iterations, "offband", sg.num_offband_pairs,
          "| oracle", so.iterations, so.termination, "%.9g" % so.final_cost, "| dxy %.2e" % np.abs(pg[:, :2] - po[:, :2]).max())
    g.close(); o.close()
