"""Development helper: device-built pair tables against host-built ones on the 50k x 400 map: same lists -> bit-identical solve."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, 50, kind=0)
a = H.Solver(); a.set_map(off, px, py, nx, ny, cov); a.set_human_constraints(rows)
a.find_correspondences(init, 3)
pa, sa = a.solve(H.MODE_JOINT, init)
lists = a.get_correspondences()
b = H.Solver(); b.set_map(off, px, py, nx, ny, cov); b.set_human_constraints(rows)
b.set_correspondences(*lists)
pb, sb = b.solve(H.MODE_JOINT, init)
print("device tables: %d it cost %.17g | host tables: %d it cost %.17g | poses identical: %s | pairs %d matches %d"
      % (sa.iterations, sa.final_cost, sb.iterations, sb.final_cost, np.array_equal(pa, pb), len(lists[0]), len(lists[3])))
