#!/usr/bin/env python3
"""Generates tests/golden/cycle_corridor.npz: one whole user correction (HitLSLAM::Run) on a small seeded corridor, stage by
stage, plus the pose covariances at the solved poses -- emitted by the INDEPENDENT mirror (mirror_cycle.py for the stages,
mirror.py for the re-solve), not by the oracle: `oracle/` is not imported here (tests/test_mirror.py asserts it).  The C oracle
and the HIP path are both held to this file by tests/test_golden.py.  It is still not reference output ("parity unpinned":
nothing reference-held exists and the reference cannot be built here).  Run from the repo root:
    python tests/golden/make_golden_cycle.py
"""
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, HERE)
sys.path.insert(0, ROOT)

import mirror as M  # noqa: E402
import mirror_cycle as MC  # noqa: E402
from hitl_slam_amd import abi  # noqa: E402  (the row dtype of the C ABI: plain numpy)
from tools import hitl_synth as S  # noqa: E402  (the synthetic map: C generator, no solver code)


def scenario(n_poses=100, obs=72, seed=20260105, first_piece=(18.5, 25.5)):
    """The corridor of tests/cycle_scenario.py without the oracle: one floor driven once; the user selects a later piece of the
    +y wall first (the observations to move) and piece 1 second, clicking on points of the drifted map."""
    p = S.params(n_poses, obs, seed=seed, poses_per_floor=n_poses)
    gt, init, cov = S.poses(p)
    off, px, py, nx, ny = S.scans(p)
    poses = np.array(init, np.float32).reshape(-1, 3)
    gt = np.array(gt, np.float64).reshape(-1, 3)
    wx, wy = MC.transform_clouds(poses, off, px, py)
    gx, gy = MC.transform_clouds(gt.astype(np.float32), off, px, py)
    pose_of = np.repeat(np.arange(n_poses), np.diff(np.asarray(off)))
    x0, y0 = gt[0, 0] - 0.15, gt[0, 1] + 0.7
    lx, ly = gx - x0, gy - y0

    def pick(pose, xa, xb):
        sel = np.where((pose_of == pose) & (np.abs(ly - 1.2) < 0.04) & (lx > xa) & (lx < xb))[0]
        assert sel.size >= 6, (pose, sel.size)
        a, b = sel[np.argmin(lx[sel])], sel[np.argmax(lx[sel])]
        return [wx[a], wy[a], wx[b], wy[b]]

    fa, fb = first_piece
    pose_first = int(round((0.5 * (fa + fb) - 0.15) / 0.3))
    pose_second = int(round((6.0 - 0.15) / 0.3))
    clicks = np.array(pick(pose_first, fa + 0.3, fb - 0.3) + pick(pose_second, 2.8, 9.2), np.float32)
    return dict(off=np.asarray(off), px=px, py=py, nx=nx, ny=ny, poses=poses, cov=np.array(cov, np.float32).reshape(-1, 9), clicks=clicks)


def main():
    sc = scenario()
    off, px, py, nx, ny, poses, cov, clicks = (sc[k] for k in ("off", "px", "py", "nx", "ny", "poses", "cov", "clicks"))
    ctype = 4
    out = dict(obs_offset=off, px=px, py=py, nx=nx, ny=ny, cov9=cov, init=poses, clicks=clicks, ctype=np.int64(ctype))
    wx, wy = MC.transform_clouds(poses, off, px, py)
    seen = MC.verify_points(wx, wy, clicks.reshape(-1, 2), 0.05)
    seg, inl, pas, its = MC.fit_segments(wx, wy, clicks.reshape(2, 4))
    counts = MC.observation_sets(wx, wy, off, seg.reshape(-1), 0.03)
    od = MC.order_user_input(counts, seg.reshape(-1))
    p2, corr, rows_l = MC.apply_explicit_correction(ctype, od["seg_xy8"], poses, od["corrected"], od["anchor"])
    p3, cov3 = MC.backprop(p2, cov, od["bounds"][0], od["bounds"][1], corr)
    p4 = MC.normalize_angles(p3)
    rows = np.zeros(len(rows_l), abi.HC_DTYPE)
    for k, r in enumerate(rows_l):
        rows[k] = (r[0], r[1], r[2], r[3], r[4], r[5], r[6])
    hc = [(int(r[0]), int(r[1]), int(r[2]), np.float32(r[3]), np.float32(r[4]), np.float32(r[5]), np.float32(r[6])) for r in rows_l]
    prob = M.Problem(M.MODE_ODOM_HC, p4, hc, None, None)
    res = M.solve(prob)
    p5 = res["poses"]
    prob5 = M.Problem(M.MODE_ODOM_HC, p5, hc, None, None)             # the problem JointOpt would build at the solved poses
    Sg = MC.pose_covariances(prob5, p5)
    out.update(seen=seen, fitted=seg, fit_inliers=inl, fit_passes=pas, fit_iterations=its, counts=counts,
               corrected=od["corrected"], anchor=od["anchor"], bounds=np.array(od["bounds"], np.int32), seg_ordered=od["seg_xy8"],
               poses_explicit=p2, correction=corr, rows=rows, poses_backprop=p3, cov_backprop=cov3, poses_normalized=p4,
               poses_solved=p5, iterations=np.int64(res["iterations"]), final_cost=np.float64(res["final_cost"]),
               termination=np.int64(res["termination"]), pose_covariances=Sg)
    path = os.path.join(HERE, "cycle_corridor.npz")
    np.savez_compressed(path, **out)
    print(path, os.path.getsize(path), "bytes; bounds", od["bounds"], "rows", rows.size, "iterations", res["iterations"],
          "fit passes", pas, "LM iterations", its, "inliers", inl)


if __name__ == "__main__":
    main()
