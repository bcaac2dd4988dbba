"""Development helper: one mode-A solve of a Problem of n poses through the fused and the level-per-launch cyclic reduction."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import hitl_slam_amd as H
from conftest import Problem
for n in [int(a) for a in sys.argv[1:]]:
    prob = Problem(n, 12, ppf=200, groups=max(1, n // 1000), kind=0)
    for rep in range(2):
        for legacy in (0, 1):
            if legacy: os.environ["HITL_CR_LEGACY"] = "1"
            else: os.environ.pop("HITL_CR_LEGACY", None)
            s = H.Solver(); prob.load(s)
            _, sm = s.solve(0, prob.init, allow_failure=True)
            print(n, "legacy" if legacy else "fused ", sm.iterations, sm.termination, "%.15g" % sm.final_cost)
            s.close()
