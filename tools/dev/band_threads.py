"""Development helper (GPU box): solve time of a 40 000-pose problem with short scans for a search window w (B = 3 w) -- the
linear solve dominates -- under the HITL_CR_THREADS setting of the environment."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import hitl_slam_amd as H
from conftest import Problem
w = int(sys.argv[1])
prob = Problem(40000, 24, ppf=200, groups=40, kind=0, window=w)
s = H.Solver(); prob.load(s)
s.find_correspondences(prob.init, w)
ts = []
for _ in range(4):
    t0 = time.perf_counter(); _, sm = s.solve(2, prob.init, allow_failure=True); ts.append((time.perf_counter() - t0) * 1e3)
print("window %d threads %s: %.2f ms, %d iterations, band %d" % (w, os.environ.get("HITL_CR_THREADS", "-"), min(ts), sm.iterations, sm.band_poses))
s.close()
