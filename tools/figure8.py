"""Stand-in for the reference's Figure8 dataset (2016-02-16-16-01-46.bag.stfs.covars is a
Google-Drive link, README.md:99-103; not in the repo, no network): a figure-8 drive through a
12 m x 8 m room with two pillars, written in the identical 16-field text format through
hitl_io_save_stfs_covars, so BASELINE configs 1 and 3 run through the real loader."""
import math

import numpy as np

ROOM = [(-6, -4, 6, -4), (6, -4, 6, 4), (6, 4, -6, 4), (-6, 4, -6, -4)]


def _box(cx, cy, h):
    return [(cx - h, cy - h, cx + h, cy - h), (cx + h, cy - h, cx + h, cy + h), (cx + h, cy + h, cx - h, cy + h),
            (cx - h, cy + h, cx - h, cy - h)]


SEGS = np.array(ROOM + _box(-3, 0, 0.6) + _box(3, 0, 0.6), float)


def raycast(x, y, th, m, rng, sigma=0.01):
    """m rays over [-135, 135] deg; returns robot-frame points and normals (float32)."""
    phi = np.radians(np.linspace(-135, 135, m))
    dx, dy = np.cos(th + phi), np.sin(th + phi)
    best = np.full(m, 1e9)
    bn = np.zeros((m, 2))
    for x0, y0, x1, y1 in SEGS:
        ex, ey = x1 - x0, y1 - y0
        den = dx * ey - dy * ex
        with np.errstate(divide="ignore", invalid="ignore"):
            t = ((x0 - x) * ey - (y0 - y) * ex) / den
            u = ((x0 - x) * dy - (y0 - y) * dx) / den
        hit = (np.abs(den) > 1e-12) & (t > 1e-6) & (u >= 0) & (u <= 1) & (t < best)
        nrm = np.array([ey, -ex]) / math.hypot(ex, ey)
        sign = np.where(dx * nrm[0] + dy * nrm[1] > 0, -1.0, 1.0)
        best = np.where(hit, t, best)
        bn[hit] = (sign[:, None] * nrm[None, :])[hit]
    r = best + sigma * rng.standard_normal(m)
    c, s = math.cos(th), math.sin(th)
    pts = np.stack([r * np.cos(phi), r * np.sin(phi)], 1)
    nrm = np.stack([c * bn[:, 0] + s * bn[:, 1], -s * bn[:, 0] + c * bn[:, 1]], 1)
    return pts.astype(np.float32), nrm.astype(np.float32)


def make(n_poses=160, m=90, seed=7, laps=2):
    """Ground truth (float64), drifted initial poses (float32), robot-frame scans."""
    rng = np.random.default_rng(seed)
    s = np.linspace(0, 2 * math.pi * laps, n_poses, endpoint=False)
    gx, gy = 3.0 * np.sin(s) * 1.0, 2.2 * np.sin(s) * np.cos(s)       # lemniscate around the two pillars
    gx = 4.2 * np.sin(s)
    gy = 2.6 * np.sin(s) * np.cos(s)
    gth = np.arctan2(np.gradient(gy), np.gradient(gx))
    gt = np.stack([gx, gy, gth], 1)
    init = np.zeros_like(gt)
    init[0] = gt[0]
    for i in range(1, n_poses):
        c, sn = math.cos(gt[i - 1, 2]), math.sin(gt[i - 1, 2])
        d = gt[i, :2] - gt[i - 1, :2]
        rel = np.array([c * d[0] + sn * d[1], -sn * d[0] + c * d[1], gt[i, 2] - gt[i - 1, 2]])
        rel[2] = math.atan2(math.sin(rel[2]), math.cos(rel[2]))
        rel += rng.normal(0, [0.004, 0.004, 0.0015])
        c, sn = math.cos(init[i - 1, 2]), math.sin(init[i - 1, 2])
        init[i] = [init[i - 1, 0] + c * rel[0] - sn * rel[1], init[i - 1, 1] + sn * rel[0] + c * rel[1], init[i - 1, 2] + rel[2]]
    pts, nrm = zip(*[raycast(gt[i, 0], gt[i, 1], gt[i, 2], m, rng) for i in range(n_poses)])
    P, N = np.concatenate(pts), np.concatenate(nrm)
    off = np.arange(n_poses + 1, dtype=np.int64) * m
    cov = np.zeros((n_poses, 9), np.float32)
    cov[:, 0] = cov[:, 4] = 0.004 ** 2 * np.arange(n_poses)
    cov[:, 8] = 0.0015 ** 2 * np.arange(n_poses)
    return dict(gt=gt, init=init.astype(np.float32), off=off, px=P[:, 0].copy(), py=P[:, 1].copy(), nx=N[:, 0].copy(),
                ny=N[:, 1].copy(), cov=cov)


def constraints(gt, n_loop=5, n_colinear=3):
    """BASELINE config 3: 5 loop-closure (type 2) + 3 colinear (type 4) groups between the two laps
    (same place revisited one lap later), rows built like AppExpCorrect::calculateConstraintTargets
    (ApplyExplicitCorrection.cpp:447-487) from ground-truth relative poses; 3 anchors x 3 corrected."""
    n = len(gt)
    half = n // 2
    rows = []
    starts = np.linspace(5, half - 10, n_loop + n_colinear).astype(int)
    for g, a0 in enumerate(starts):
        ctype = 2 if g < n_loop else 4
        for a in range(a0, a0 + 3):
            rel_pen = math.atan2(math.sin(0.0 - gt[a, 2]), math.cos(0.0 - gt[a, 2])) + math.pi / 2
            for c in range(a0 + half, a0 + half + 3):
                d = gt[c, :2] - gt[a, :2]
                ca, sa = math.cos(gt[a, 2]), math.sin(gt[a, 2])
                da = gt[c, 2] - gt[a, 2]
                rows.append((ctype, c, a, ca * d[0] + sa * d[1], -sa * d[0] + ca * d[1], math.atan2(math.sin(da), math.cos(da)), rel_pen))
    return rows
