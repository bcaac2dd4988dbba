"""Development: the persistent PCG launch's recurrence residual against the true residual b - A x of what it returns (HITL_TIMING=1 prints
both per LM iteration), on the c3 map (mode C) and on an ill-conditioned problem (mode B = scan-to-scan factors only, all pairs: the
oracle's LM runs to its 100-iteration cap there)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import figure8
import hitl_slam_amd as H
from hitl_slam_amd import abi
from hitl_slam_amd.api import hc_rows
for n, m, laps, mode in ((1000, 360, 4, abi.MODE_JOINT), (180, 90, 2, abi.MODE_STF_ONLY)):
    d = figure8.make(n, m, seed=7 if n == 1000 else 13, laps=laps)
    rows = hc_rows(figure8.constraints(d["gt"], 5, 3))
    s = H.Solver(H.default_options())
    s.set_map(d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"]); s.set_human_constraints(rows)
    s.find_correspondences(d["init"], 0)
    sys.stderr.write("=== %d x %d mode %d\n" % (n, m, mode)); sys.stderr.flush()
    p, sm = s.solve(mode, d["init"], allow_failure=True)
    sys.stderr.write("=== %d LM iterations, cost %.12g\n" % (sm.iterations, sm.final_cost))
    s.close()
