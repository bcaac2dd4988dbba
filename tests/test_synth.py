"""The synthetic-map generator is bit-reproducible (own RNG) and produces sane scans."""
import hashlib

import numpy as np

from tools import hitl_synth as S


def _digest(*arrs):
    h = hashlib.sha256()
    for a in arrs:
        h.update(np.ascontiguousarray(a).tobytes())
    return h.hexdigest()


def test_deterministic_and_range_splittable():
    p = S.params(400, 64, seed=123, poses_per_floor=100)
    gt, init, cov = S.poses(p)
    full = S.scans(p)
    assert _digest(gt, init, cov, *full) == _digest(*S.poses(p), *S.scans(p))
    part = S.scans(p, 100, 300)
    for a, b in zip(full[1:], part[1:]):
        assert np.array_equal(a[100 * 64:300 * 64], b)         # a rank can generate just its submap
    p2 = S.params(400, 64, seed=124, poses_per_floor=100)
    assert _digest(*S.scans(p2)) != _digest(*full)


def test_scans_are_sane():
    p = S.params(200, 90, poses_per_floor=200)
    gt, init, cov = S.poses(p)
    off, px, py, nx, ny = S.scans(p)
    assert off[-1] == 200 * 90 and np.all(np.diff(off) == 90)       # exactly M observations per pose
    r = np.hypot(px, py)
    assert r.min() > 0.05 and r.max() < 80.0                         # bounded sight lines
    assert np.allclose(np.hypot(nx, ny), 1.0, atol=1e-6)
    assert np.all(px * nx + py * ny < 0.05)                          # normals face the sensor
    assert np.abs(gt[:, :2]).max() < 600                             # compact layout keeps fp32 resolution
    assert np.abs(init - gt).max() < 1.0


def test_constraint_rows_follow_the_reference_generator():
    p = S.params(300, 16, poses_per_floor=100)
    gt, _, _ = S.poses(p)
    rows, n = S.constraints(p, gt, 3, kind=0)
    assert n == 75
    types = sorted({rows[k].type for k in range(n)})
    assert types == [2, 4, 5]                                        # mixed round-robin over 3 groups
    for k in range(n):
        a, c = rows[k].anchor_pose_id, rows[k].constrained_pose_id
        assert 0 < a < 100 and 200 <= c < 300                       # first third -> last third
        # ApplyExplicitCorrection.cpp:465-474: deltas are the relative pose in the anchor frame
        ca, sa = np.cos(gt[a, 2]), np.sin(gt[a, 2])
        d = gt[c, :2] - gt[a, :2]
        assert abs(rows[k].delta_parallel - (ca * d[0] + sa * d[1])) < 1e-4
        assert abs(rows[k].delta_perpendicular - (-sa * d[0] + ca * d[1])) < 1e-4
