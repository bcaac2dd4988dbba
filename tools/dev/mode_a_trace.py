import sys
sys.path.insert(0, "/root/repo")
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, 50, kind=0)
s = H.Solver(); s.set_map(off, px, py, nx, ny, cov); s.set_human_constraints(rows)
for _ in range(3):
    out, sm = s.solve(0, init)
print("mode A:", sm.iterations, sm.total_ms)
