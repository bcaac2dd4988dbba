import sys, time, os
sys.path.insert(0, "/root/repo"); sys.path.insert(0, "/root/repo/tests")
import numpy as np
from conftest import Problem
os.environ.setdefault("OMP_NUM_THREADS", "16")
import hitl_slam_amd as H
from hitl_slam_amd import abi
from oracle import hitl_oracle as O
T = {}
def tick(name, t0): T[name] = T.get(name, 0.0) + time.time() - t0
for rep in range(2):
    t0 = time.time(); prob = Problem(104, 156, seed=20260405, ppf=77, groups=1, kind=2, window=2); tick("problem", t0)
    t0 = time.time(); g = H.Solver(); tick("gpu create", t0)
    o = O.OracleSolver()
    t0 = time.time(); prob.load(g); tick("gpu load", t0)
    t0 = time.time(); prob.load(o); tick("orc load", t0)
    t0 = time.time(); g.find_correspondences(prob.init, 2); tick("gpu corr", t0)
    t0 = time.time(); o.find_correspondences(prob.init, 2, 0); tick("orc corr", t0)
    for mode in (0, 2):
        t0 = time.time(); pg, sg = g.solve(mode, prob.init); tick("gpu solve", t0)
        t0 = time.time(); po, so = o.solve(mode, prob.init); tick("orc solve", t0)
        t0 = time.time(); Sg = g.pose_covariances(mode, po); tick("gpu cov", t0)
        t0 = time.time(); So = o.pose_covariances(mode, po); tick("orc cov", t0)
    t0 = time.time(); g.close(); o.close(); tick("close", t0)
    print(rep, {k: round(v, 3) for k, v in T.items()}); T.clear()
print("OMP_NUM_THREADS", os.environ.get("OMP_NUM_THREADS"), "cpus", len(os.sched_getaffinity(0)))
