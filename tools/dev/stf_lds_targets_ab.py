"""VERDICT r5 item 3, the counters route: what would target scans resident in LDS buy k_linearize_stf_r?  A/B on the C4 map between the
shipped kernel and its test-hook instantiation that reads every target point from LDS (ABL = 1: the fp32 copy of the SOURCE scan at
the target's index -- wrong numbers on purpose, the instruction stream of an LDS-resident kernel with none of its staging cost).
STF_LDS=0|1 picks one (for the rocprofv3 --pmc passes of tools/dev/pmc_stf_ab.sh); unset: both, HIP-event time per launch."""
import os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = int(os.environ.get("POSES", 50000)), 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, max(1, n // 1000), kind=4)
which = [int(x) for x in os.environ.get("STF_LDS", "0,1,0,1").split(",")]
o = H.default_options(); o.timing_level = 2
o.max_iterations = 12                    # (the ablated kernel's numbers are wrong: a fixed number of launches, whatever the LM makes of them)
corr = None
for w in which:
    H.set_test_switch("stf_lds_targets", w)          # (read when a solver is created)
    s = H.Solver(o)
    s.set_map(off, px, py, nx, ny, cov)
    s.set_human_constraints(rows)
    s.find_correspondences(init, 3)
    s.solve(H.MODE_JOINT, init, allow_failure=True)
    best = None
    for _ in range(int(os.environ.get("REPS", 3))):
        _, sm = s.solve(H.MODE_JOINT, init, allow_failure=True)
        v = 1e3 * sm.linearize_ms / max(sm.linearize_timed_calls, 1)
        best = v if best is None else min(best, v)
    print("stf_lds_targets=%d: k_linearize_stf_r %.1f us per launch (%d launches per solve, final cost %.6g)" % (w, best, sm.linearize_timed_calls, sm.final_cost))
    s.close()
