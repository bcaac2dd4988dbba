"""Development helper: wall clock of C4 (or POSES=...) re-solves with the host-driven and the device-driven LM loop (test hook device_lm), modes A and C."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = int(os.environ.get("POSES", 50000)), int(os.environ.get("OBS", 400))
p = S.params(n, m, seed=20264102, poses_per_floor=200 if n >= 400 else n // 2)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, max(1, n // 1000), kind=4)
corr = None
res = {}
VARIANTS = {"lm": (("host", dict(device_lm=0)), ("device", dict(device_lm=1)), ("device ahead 1", dict(device_lm=1, lm_ahead=1)), ("device ahead 6", dict(device_lm=1, lm_ahead=6))),
            "tail": (("tail 4 launches", dict(cr_tail=0)), ("tail fused", dict(cr_tail=1)), ("fused + device lm", dict(cr_tail=1, device_lm=1)), ("4 launches + device lm", dict(cr_tail=0, device_lm=1)))}
its = {}
for rep in range(max(1, int(os.environ.get("REPS", 2)))):
    for tag, hooks in VARIANTS[os.environ.get("AB", "lm")]:
        H.set_test_switch(None)
        for k, v in hooks.items():
            H.set_test_switch(k, v)
        s = H.Solver()
        s.set_map(off, px, py, nx, ny, cov)
        s.set_human_constraints(rows)
        if corr is None:
            s.find_correspondences(init, 3); corr = s.get_correspondences()
        else:
            s.set_correspondences(*corr)
        for mode in (H.MODE_ODOM_HC, H.MODE_JOINT):
            s.solve(mode, init)
            ts = []
            for _ in range(5):
                t0 = time.perf_counter(); po, sm = s.solve(mode, init); ts.append((time.perf_counter() - t0) * 1e3)
            res.setdefault((tag, mode), []).append(min(ts))
            its[(tag, mode)] = (sm.iterations, sm.final_cost)
        s.close()
# the minimum over the repetitions (variants interleaved: box drift hits all of them alike)
for tag, _ in VARIANTS[os.environ.get("AB", "lm")]:
    out = []
    for mode in (H.MODE_ODOM_HC, H.MODE_JOINT):
        t, (it, cost) = min(res[(tag, mode)]), its[(tag, mode)]
        out.append("mode %d: %.3f ms (%d it, cost %.15g, %.1f us/it)" % (mode, t, it, cost, 1e3 * t / max(it, 1)))
    print("%-24s %s" % (tag, " | ".join(out)), flush=True)
