"""Development helper: the round-2/3 reference point of the PCG path -- ONE straight corridor searched all pairs (every pose sees most of the
others: the densest off-band rows), mode C re-solve time.  usage: corridor_allpairs.py [poses] [obs]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import hitl_slam_amd as H
from conftest import Problem
n = int(sys.argv[1]) if len(sys.argv) > 1 else 1000
m = int(sys.argv[2]) if len(sys.argv) > 2 else 360
prob = Problem(n, m, ppf=n, groups=2)
for fused in (1, 0):
    H.set_test_switch(None); H.set_test_switch("pcg_fused", fused)
    opt = H.default_options(); opt.timing_level = 2
    g = prob.load(H.Solver(opt))
    g.find_correspondences(prob.init, 0)
    g.solve(H.MODE_JOINT, prob.init, allow_failure=True)
    ts = []
    for _ in range(3):
        t0 = time.perf_counter(); p, sm = g.solve(H.MODE_JOINT, prob.init, allow_failure=True); ts.append((time.perf_counter() - t0) * 1e3)
    cg = sm.linear_iterations - sm.iterations
    print("corridor %d x %d all pairs, %s: %.1f ms per re-solve, %d LM / %d CG iterations, %d off-band block pairs (%.0f per pose), %.3f ms linear solve per LM iteration, %s us per CG iteration, cost %.12g"
          % (n, m, "one persistent PCG launch per solve" if fused else "launch-per-stage PCG loop (rounds 2-5)", min(ts), sm.iterations, cg, sm.num_offband_pairs,
             sm.num_offband_pairs / n, sm.linear_solver_ms / max(sm.iterations, 1), ("%.1f" % (1e3 * sm.pcg_ms / sm.pcg_timed_iterations)) if sm.pcg_timed_iterations else "-", sm.final_cost), flush=True)
    g.close()
