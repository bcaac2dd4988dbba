#!/usr/bin/env python3
"""Generates tests/golden/*.npz from tests/golden/mirror.py -- the independent numpy restatement of the reference's path.

Until round 2 these fixtures were the C oracle's own output (a regression pin).  Since round 3 they come from the mirror,
which shares no code with `oracle/hitl_oracle.c` or the product, and tests/test_golden.py holds BOTH the C oracle and the
HIP path to them.  They are still not reference outputs: the reference has no golden vectors and cannot be built or
imported here, so parity stays "unpinned" by the letter -- but two readings of the same reference text by
different routes (literal autodiff functors + dense normal equations vs. analytic Jacobians + banded Cholesky) now have to
agree before anything is called a target.

This script does not import `oracle/`.  Inputs come from the synthetic-map tool (tools/hitl_synth, a data generator).
Run from the repo root:   python tests/golden/make_golden.py
"""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))

from conftest import Problem  # noqa: E402   (inputs only: wraps tools/hitl_synth)
import mirror as M  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))

CASES = {
    # name: (poses, obs/pose, poses/floor, constraint groups, kind, window, seed)
    "corridor_colinear": (24, 48, 24, 1, 4, 3, 20260102),
    "two_floors_mixed": (30, 40, 15, 2, 0, 2, 20260103),
    "all_pairs_small": (12, 32, 12, 1, 2, 0, 20260104),
}


def rows_to_array(rows, n):
    a = np.zeros((n, 7), np.float64)
    for k in range(n):
        r = rows[k]
        a[k] = [r.type, r.constrained_pose_id, r.anchor_pose_id, r.delta_parallel, r.delta_perpendicular,
                r.delta_angle, r.relative_penalty_dir]
    return a


def main():
    for name, (n, m, ppf, groups, kind, window, seed) in CASES.items():
        t0 = time.time()
        prob = Problem(n, m, seed=seed, ppf=ppf, groups=groups, kind=kind, window=window)
        hc = rows_to_array(*prob.rows)
        scans = (prob.off, prob.px, prob.py, prob.nx, prob.ny)
        # FindSTFCorrespondences: the KD-tree walk (the reference's; the product's default) and the brute-force definition
        kd = M.find_stf_correspondences(prob.init, *scans, window, 1)
        bf = M.find_stf_correspondences(prob.init, *scans, window, 0)
        out = dict(obs_offset=prob.off, px=prob.px, py=prob.py, nx=prob.nx, ny=prob.ny, cov9=prob.cov,
                   init=prob.init, hc=hc, window=np.int64(window),
                   pair_src=kd[0], pair_tgt=kd[1], match_offset=kd[2], src_idx=kd[3].astype(np.int16), tgt_idx=kd[4].astype(np.int16),
                   bf_pair_src=bf[0], bf_pair_tgt=bf[1], bf_match_offset=bf[2], bf_src_idx=bf[3].astype(np.int16),
                   bf_tgt_idx=bf[4].astype(np.int16))
        rng = np.random.default_rng(seed)
        xlin = prob.init.astype(np.float64) + 0.01 * rng.standard_normal(prob.init.shape)
        out["xlin"] = xlin
        for mode in (0, 1, 2):
            P = M.Problem(mode, prob.init, hc, kd, scans)
            cost, g, pr, pj = M.linearize(P, xlin)
            out["lin_cost_%d" % mode] = np.float64(cost)
            out["lin_grad_%d" % mode] = g
            if mode:
                out["lin_pair_r_%d" % mode] = pr
                out["lin_pair_J_%d" % mode] = pj
            for late, prefix in ((False, "solve"), (True, "late_solve")):
                s = M.solve(P, tolerances_after_accept=late)
                out["%s_poses_%d" % (prefix, mode)] = s["poses"]
                out["%s_x64_%d" % (prefix, mode)] = s["x"]
                out["%s_summary_%d" % (prefix, mode)] = np.array([s["termination"], s["iterations"], s["successful_steps"],
                                                               s["num_residuals"], s["num_hc_residuals"]], np.int64)
                out["%s_cost_%d" % (prefix, mode)] = np.array([s["initial_cost"], s["final_cost"]])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: %d poses, %d pairs / %d matches (KD walk), %d / %d (brute force), %d hc rows -> %.1f KB, %.0f s; "
              "iterations by mode %s, with the tolerances after the accept %s" %
              (name, n, len(kd[0]), len(kd[3]), len(bf[0]), len(bf[3]), prob.rows[1], os.path.getsize(path) / 1024.0, time.time() - t0,
               [int(out["solve_summary_%d" % k][1]) for k in range(3)], [int(out["late_solve_summary_%d" % k][1]) for k in range(3)]))


if __name__ == "__main__":
    main()
