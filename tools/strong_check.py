"""Development helper: the SAME problem bench.py --scaling strong --gpus 2 hands to two submap ranks (50k x 400 map, the two
ranks' constraint rows), solved on ONE GPU: iteration count / termination / cost to hold the 2-rank result against."""
import os, sys, ctypes as C
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import hitl_slam_amd as H
from hitl_slam_amd import abi
from tools import hitl_synth as S
n, m = int(sys.argv[1]) if len(sys.argv) > 1 else 50000, 400
world = int(sys.argv[2]) if len(sys.argv) > 2 else 2
ppf, window = 200, 3
groups = max(1, int(round(50 * n / 50000)))          # bench.py --poses keeps the constraint density of the named workload
p = S.params(n, m, seed=20260102 + 4000, poses_per_floor=ppf)
gt, init, cov = S.poses(p)
cuts = [0] + [int(round(r * n / world / ppf)) * ppf for r in range(1, world)] + [n]
g = max(1, groups // world)
parts = [S.constraints(p, gt, g, kind=0, begin=cuts[r], end=cuts[r + 1]) for r in range(world)]
tot = sum(k for _, k in parts)
rows = (abi.HumanConstraint * tot)()
w = 0
for arr, k in parts:
    for q in range(k):
        rows[w] = arr[q]; w += 1
off, px, py, nx, ny = S.scans(p)
s = H.Solver(); s.set_map(off, px, py, nx, ny, cov); s.set_human_constraints((rows, tot))
s.find_correspondences(init, window)
out, summ = s.solve(H.MODE_JOINT, init, allow_failure=True)
print("single GPU on the strong-%d problem:" % world + " iterations %d, termination %d, cost %.9g -> %.9g, rows %d" % (summ.iterations, summ.termination, summ.initial_cost, summ.final_cost, tot))
