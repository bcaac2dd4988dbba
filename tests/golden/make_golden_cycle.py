#!/usr/bin/env python3
"""Generates tests/golden/cycle_corridor.npz: one whole user correction (HitLSLAM::Run) on a small seeded corridor, with
the oracle's result after every stage, plus the pose covariances at the solved poses.  Like the other fixtures this pins
the ORACLE and gives the HIP path a fixed target; it is not reference output ("parity unpinned").  Run from the repo root:
    python tests/golden/make_golden_cycle.py
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)

from hitl_slam_amd import abi  # noqa: E402
from oracle import hitl_oracle as O  # noqa: E402
from tests.cycle_scenario import Scenario  # noqa: E402

OUT = os.path.dirname(os.path.abspath(__file__))


def main():
    sc = Scenario(n_poses=100, obs=72, seed=20260105)
    pr = sc.prob
    o = O.OracleSolver()
    pr.load(o)
    out = dict(obs_offset=pr.off, px=pr.px, py=pr.py, nx=pr.nx, ny=pr.ny, cov9=sc.cov, init=sc.poses, clicks=sc.selected,
               ctype=np.int64(4))
    seen = o.verify_points(sc.poses, sc.selected.reshape(-1, 2), 0.05)
    seg, inl, pas, its = o.fit_segments(sc.poses, sc.selected.reshape(2, 4))
    counts, _ = o.observation_sets(sc.poses, seg.reshape(-1), 0.03, want_mask=False)
    od = o.order_user_input(counts, seg.reshape(-1))
    p2, corr, rows = o.apply_explicit_correction(4, od["seg_xy8"], sc.poses, od["corrected"], od["anchor"])
    p3, cov3 = o.backprop(p2, sc.cov, od["bounds"][0], od["bounds"][1], corr)
    p4 = o.normalize_angles(p3)
    o.set_human_constraints(rows)
    p5, summ = o.solve(abi.MODE_ODOM_HC, p4)
    S = o.pose_covariances(abi.MODE_ODOM_HC, p5)
    out.update(seen=seen, fitted=seg, fit_inliers=inl, fit_passes=pas, fit_iterations=its, counts=counts,
               corrected=od["corrected"], anchor=od["anchor"], bounds=np.array(od["bounds"], np.int32), seg_ordered=od["seg_xy8"],
               poses_explicit=p2, correction=corr, rows=rows, poses_backprop=p3, cov_backprop=cov3, poses_normalized=p4,
               poses_solved=p5, iterations=np.int64(summ.iterations), final_cost=np.float64(summ.final_cost),
               termination=np.int64(summ.termination), pose_covariances=S)
    path = os.path.join(OUT, "cycle_corridor.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; bounds", od["bounds"], "rows", rows.size, "iterations", summ.iterations)


if __name__ == "__main__":
    main()
