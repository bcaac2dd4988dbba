"""Development helper: all-pairs correspondences on one corridor (hundreds of poses, most pose pairs off the band) --
GPU (off-band PCG) against the oracle's exact solve.  usage: allpairs_vs_oracle.py [poses] [obs]"""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hitl_slam_amd as H
from oracle import hitl_oracle as O
from conftest import Problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 300
m = int(sys.argv[2]) if len(sys.argv) > 2 else 120
prob = Problem(n, m, ppf=n, groups=2)
g, o = H.Solver(), O.OracleSolver()
prob.load(g); prob.load(o)
t0 = time.time(); g.find_correspondences(prob.init, 0); tg = time.time() - t0
t0 = time.time(); o.find_correspondences(prob.init, 0, 0); to = time.time() - t0
same = all(np.array_equal(a, b) for a, b in zip(g.get_correspondences(), o.get_correspondences()))
print("search: gpu %.1f ms, oracle %.1f s, identical sets: %s, pairs %d" % (tg * 1e3, to, same, len(g.get_correspondences()[0])))
t0 = time.time(); pg, sg = g.solve(2, prob.init, allow_failure=True); tg = time.time() - t0
t0 = time.time(); po, so = o.solve(2, prob.init); to = time.time() - t0
print("solve: gpu %.1f ms (%d LM, %d CG, term %d, cost %.9g, offband %d) | oracle %.1f s (%d LM, term %d, cost %.9g) | dxy %.2e dth %.2e"
      % (tg * 1e3, sg.iterations, sg.linear_iterations, sg.termination, sg.final_cost, sg.num_offband_pairs, to, so.iterations, so.termination, so.final_cost,
         np.abs(pg[:, :2] - po[:, :2]).max(), np.abs(pg[:, 2] - po[:, 2]).max()))
