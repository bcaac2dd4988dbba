#!/usr/bin/env python3
"""Generates tests/golden/*.npz: small seeded inputs and the oracle's outputs for them.

The reference has no golden vectors and cannot be built or imported here, so
these fixtures pin the ORACLE (regression) and give the GPU tests a fixed target; they are
not reference outputs ("parity unpinned").  The oracle's optimum is cross-checked against
scipy.optimize.least_squares in tests/test_oracle_solve.py.  Run from the repo root:
    python tests/golden/make_golden.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

from conftest import Problem  # noqa: E402
from hitl_slam_amd import abi  # noqa: E402
from oracle import hitl_oracle as O  # noqa: E402

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
        prob = Problem(n, m, seed=seed, ppf=ppf, groups=groups, kind=kind, window=window)
        o = prob.load(O.OracleSolver())
        o.find_correspondences(prob.init, window, 0)
        ps, pt, mo, i0, i1 = o.get_correspondences()
        out = dict(obs_offset=prob.off, px=prob.px, py=prob.py, nx=prob.nx, ny=prob.ny, cov9=prob.cov,
                   init=prob.init, hc=rows_to_array(*prob.rows), window=np.int64(window),
                   pair_src=ps, pair_tgt=pt, match_offset=mo, src_idx=i0.astype(np.int16), tgt_idx=i1.astype(np.int16))
        rng = np.random.default_rng(seed)
        xlin = prob.init.astype(np.float64) + 0.01 * rng.standard_normal(prob.init.shape)
        out["xlin"] = xlin
        for mode in (0, 1, 2):
            cost, g, pr, pj = o.linearize(mode, prob.init, xlin)
            poses, summ = o.solve(mode, prob.init)
            x64, _ = o.solve_f64(mode, prob.init)
            out["lin_cost_%d" % mode] = np.float64(cost)
            out["lin_grad_%d" % mode] = g
            if mode:
                out["lin_pair_r_%d" % mode] = pr
                out["lin_pair_J_%d" % mode] = pj
            out["solve_poses_%d" % mode] = poses
            out["solve_x64_%d" % mode] = x64
            out["solve_summary_%d" % mode] = np.array([summ.termination, summ.iterations, summ.successful_steps,
                                                      summ.num_residuals, summ.num_hc_residuals], np.int64)
            out["solve_cost_%d" % mode] = np.array([summ.initial_cost, summ.final_cost])
        path = os.path.join(OUT, name + ".npz")
        np.savez_compressed(path, **out)
        print("%s: %d poses, %d pairs, %d matches, %d hc rows -> %.1f KB" %
              (name, n, len(ps), len(i0), prob.rows[1], os.path.getsize(path) / 1024.0))


if __name__ == "__main__":
    main()
