"""Development helper: the C4 search + 3 re-solves (the LM path's own kernels) for rocprofv3 --kernel-trace A/B runs."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
from hitl_slam_amd.api import apply_test_switches_from_env
apply_test_switches_from_env()          # HITL_TEST_SWITCHES="cr_seq=0": the library's test hooks, set by this script
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

if os.environ.get("SEQ_PROBE"):         # development build with -DHITL_CR_PROBE only: the workgroup timeline of the LAST k_cr_seq launch
    import ctypes as C
    from hitl_slam_amd.api import load_library
    nwg = 1042
    buf = (C.c_ulonglong * (4 * nwg))()
    load_library().hitl_seq_probe_dump(buf, nwg)
    a = np.frombuffer(buf, dtype=np.uint64).reshape(-1, 4).astype(np.int64)
    t0 = a[:, 0].min()
    st, pro, en = (a[:, 0] - t0) / 100.0, (a[:, 1] - a[:, 0]) / 100.0, (a[:, 2] - t0) / 100.0      # microseconds
    print("k_cr_seq timeline (us): kernel span %.1f; starts: min %.1f median %.1f p90 %.1f max %.1f; staging median %.1f; lifetime median %.1f p10 %.1f p90 %.1f max %.1f"
          % (en.max(), st.min(), np.median(st), np.percentile(st, 90), st.max(), np.median(pro), np.median(en - st), np.percentile(en - st, 10), np.percentile(en - st, 90), (en - st).max()))
    print("start-time histogram (2 us bins):", np.histogram(st, bins=np.arange(0, st.max() + 2, 2))[0].tolist())
    print("end-time histogram (2 us bins):", np.histogram(en, bins=np.arange(0, en.max() + 2, 2))[0].tolist())
    hw = a[:, 3] & 0xffffffff
    xcc = (a[:, 3] >> 32) & 0xf
    cu = (hw >> 8) & 0xf; sh = (hw >> 12) & 0x1; se = (hw >> 13) & 0x7
    key = xcc * 1000 + se * 100 + sh * 10 + cu
    u, cnt = np.unique(key, return_counts=True)
    print("distinct (xcc, se, sh, cu):", len(u), "workgroups per CU: min %d max %d; histogram" % (cnt.min(), cnt.max()), np.bincount(cnt).tolist())
    late = st > np.median(st) + 5
    print("late starters:", int(late.sum()), "their lifetimes median %.1f" % (np.median((en - st)[late]) if late.any() else 0.0))
