"""Development helper: does the ORDER of a pair's matches matter to k_linearize_stf_r?  The search emits them in
source-point order; the gathers of a half-wave then touch whatever target lines those 32 sources matched.  ORDER=src|tgt|rand picks
the order handed to set_correspondences (sums are order-independent up to rounding); all three when unset.  Prints the HIP-event time
per launch (timing_level 2: every launch bracketed)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import hitl_slam_amd as H
from tools import hitl_synth as S
n, m = int(os.environ.get("POSES", 50000)), 400
p = S.params(n, m, seed=20264102, poses_per_floor=200)
gt, init, cov = S.poses(p)
off, px, py, nx, ny = S.scans(p)
rows = S.constraints(p, gt, max(1, n // 1000), kind=4)
o = H.default_options(); o.timing_level = 2
s = H.Solver(o)
s.set_map(off, px, py, nx, ny, cov)
s.set_human_constraints(rows)
s.find_correspondences(init, 3)
ps, pt, mo, i0, i1 = s.get_correspondences()
pair_of = np.repeat(np.arange(len(ps)), np.diff(mo))
orders = os.environ.get("ORDER", "src,tgt,rand").split(",")
rng = np.random.default_rng(1)
for order in orders:
    if order == "src": perm = np.arange(len(i0))
    elif order == "tgt": perm = np.lexsort((i0, i1, pair_of))
    else: perm = np.lexsort((rng.random(len(i0)), pair_of))
    s.set_correspondences(ps, pt, mo, i0[perm], i1[perm])
    s.solve(H.MODE_JOINT, init)
    best = None
    for _ in range(int(os.environ.get("REPS", 3))):
        _, sm = s.solve(H.MODE_JOINT, init)
        v = 1e3 * sm.linearize_ms / max(sm.linearize_timed_calls, 1)
        best = v if best is None else min(best, v)
    # distinct 128-byte target lines per half-wave trip (32 lanes x 2 matches = 64 consecutive matches of a pair), averaged over a sample of pairs
    smp = rng.choice(len(ps), 2000, replace=False)
    lines = []
    for q in smp:
        t = i1[perm][mo[q]:mo[q + 1]].astype(np.int64) * 16 // 128
        for a in range(0, len(t), 64):
            lines.append(len(np.unique(t[a:a + 32])) + (len(np.unique(t[a + 32:a + 64])) if a + 32 < len(t) else 0))
    print("order %-4s: k_linearize_stf %.1f us per launch, %d iterations, cost %.12g; distinct 128-B target lines per trip (2 gather instructions): %.1f"
          % (order, best, sm.iterations, sm.final_cost, float(np.mean(lines))), flush=True)
s.close()
