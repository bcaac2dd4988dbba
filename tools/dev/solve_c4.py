"""Development helper: the C4 search + 3 re-solves (the LM path's own kernels) for rocprofv3 --kernel-trace A/B runs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = 50000, 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
opt = H.default_options()
if os.environ.get("HITL_FORCE_RCCL"):           # one-rank RCCL communicator: the exchanges of the submap path really go through ncclAllGather
    import ctypes as C
    from hitl_slam_amd.api import comm_unique_id
    _uid = (C.c_char * 128).from_buffer_copy(comm_unique_id())
    opt.rccl_unique_id = C.cast(_uid, C.c_void_p)
s = H.Solver(opt)
s.set_map(off, px, py, nx, ny, cov)
s.set_human_constraints(S.constraints(p, gt, 50, kind=4))
s.find_correspondences(init, 3)
ts = []
for _ in range(int(os.environ.get("REPS", "4"))):
    t0 = time.perf_counter()
    poses, summ = s.solve(H.MODE_JOINT, init)
    ts.append((time.perf_counter() - t0) * 1e3)
print("solve ms", " ".join("%.2f" % t for t in ts), "iterations", summ.iterations, "cost %.15g" % summ.final_cost, "submap form", int(summ.submap_linear_solver))

if os.environ.get("CR_PROBE"):          # development build with -DHITL_CR_PROBE only
    from hitl_slam_amd.api import load_library
    load_library().hitl_cr_probe_dump()
s.close()
