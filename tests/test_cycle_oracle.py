"""CPU checks of the correction-cycle restatement (oracle/hitl_oracle.c, last section) and of the host index logic
of the product (hitl_order_user_input needs no GPU).  References: EMinput.cpp:153-455,
ApplyExplicitCorrection.cpp:150-494, Backprop.cpp:98-213, HitLSLAM.cpp:380-480."""
import ctypes as C

import numpy as np
import pytest

from hitl_slam_amd import abi
from oracle import hitl_oracle as O
from tests.cycle_scenario import Scenario


@pytest.fixture(scope="module")
def sc():
    return Scenario()


@pytest.fixture(scope="module")
def orc(sc):
    o = O.OracleSolver()
    sc.prob.load(o)
    yield o
    o.close()


def _chain(n, seed=3):
    rng = np.random.default_rng(seed)
    th = np.cumsum(rng.normal(0, 0.05, n))
    xy = np.cumsum(np.c_[0.3 * np.cos(th), 0.3 * np.sin(th)], axis=0) + [12.0, -7.0]
    poses = np.c_[xy, th].astype(np.float32)
    cov = np.zeros((n, 9), np.float32)
    k = np.arange(n)
    cov[:, 0] = cov[:, 4] = 1e-5 * (1 + k % 37)
    cov[:, 8] = 2e-7 * (1 + k % 23)
    cov[:, 1] = cov[:, 3] = 1e-6
    cov[:, 2] = cov[:, 6] = 3e-7
    cov[:, 5] = cov[:, 7] = 2e-7
    return poses, cov


def test_backprop_closed_form_and_rounding_noise():
    """The O(span^2) loops equal a prefix sum: with Theta_j = R(sum_{i<=j} dtheta_i) and
    t_j = t_{j-1} + (Theta_{j-1} - Theta_j) p_j, pose k ends at Theta_{k-1} p_k + t_{k-1}.  fp64 literal == closed form;
    fp32 literal differs from both by its own rounding noise only."""
    n, lo, hi = 400, 37, 352
    poses, cov = _chain(n)
    corr = np.array([0.8, -0.5, 0.12], np.float32)
    p32, cov_out, p64 = O.backprop_f64(poses, cov, lo, hi, corr)
    rs = cov[:, 8].astype(np.float64)
    ts = ((cov[:, 0] + cov[:, 4]).astype(np.float32)).astype(np.float64) / 2.0
    sr = np.float32(0)
    st = np.float32(0)
    for i in range(lo, hi + 1):
        sr = np.float32(sr + cov[i, 8])
        st = np.float32(st + np.float32(ts[i]))
    sr = float(np.float32(sr + np.float32(0.0001)))
    st = float(np.float32(st + np.float32(0.001)))
    x = poses.astype(np.float64)
    dth = rs[lo:hi] / sr * float(corr[2])
    cum = np.cumsum(dth)
    exp = x.copy()
    t = np.zeros(2)
    R = lambda a: np.array([[np.cos(a), -np.sin(a)], [np.sin(a), np.cos(a)]])
    prev = np.eye(2)
    for q in range(hi - lo):             # pivot q
        cur = R(cum[q])
        t = t + (prev - cur) @ x[lo + q, :2]
        exp[lo + q + 1, :2] = cur @ x[lo + q + 1, :2] + t
        prev = cur
    exp[lo:hi, 2] += cum
    exp[hi, 2] += cum[-1]
    dest = poses[hi, :2].astype(np.float32) + corr[:2]
    trans = dest.astype(np.float64) - exp[hi, :2]
    w = np.concatenate([[0.0], np.cumsum(ts[lo:hi] / st)])
    exp[lo:hi + 1, :2] += w[:, None] * trans[None, :]
    assert np.abs(p64 - exp).max() < 1e-9
    # fp32 noise of ~span sequential updates at |x| ~ 100 m: already above 1e-4 m here, which is why the product
    # reproduces the updates one by one instead of evaluating the closed form
    assert np.abs(p32 - p64).max() < 2e-3
    assert np.array_equal(p32[:lo], poses[:lo]) and np.array_equal(p32[hi + 1:], poses[hi + 1:])
    # covariance shrink: (0,2) twice, (1,2) never, rows outside [lo, hi) untouched
    rb = np.float32(1) / (np.float32(1) + cov[hi - 1, 8] / np.float32(0.0001))
    tb = np.float32(1) / (np.float32(1) + np.float32(ts[hi - 1]) / np.float32(0.001))
    assert np.array_equal(cov_out[lo:hi, 0], cov[lo:hi, 0] * tb)
    assert np.array_equal(cov_out[lo:hi, 2], (cov[lo:hi, 2] * rb) * rb)
    assert np.array_equal(cov_out[lo:hi, 5], cov[lo:hi, 5])
    assert np.array_equal(cov_out[lo:hi, 8], cov[lo:hi, 8] * rb)
    assert np.array_equal(cov_out[hi:], cov[hi:]) and np.array_equal(cov_out[:lo], cov[:lo])


def test_backprop_needs_room():
    poses, cov = _chain(20)
    o = O.OracleSolver()
    p, c = o.backprop(poses, cov, 7, 7, [1, 1, 0.1])     # "No poses between bounds"
    assert np.array_equal(p, poses) and np.array_equal(c, cov)
    p, c = o.backprop(poses, cov, 9, 3, [1, 1, 0.1])
    assert np.array_equal(p, poses)
    o.close()


@pytest.mark.parametrize("ctype", [2, 4, 5, 6])
def test_explicit_correction_rows_reproduce_the_moved_poses(orc, sc, ctype):
    """Rows written by calculateConstraintTargets and read by AddHumanConstraints (JointOptimization.cpp:980-989) must
    agree: the target of every new row is the corrected pose it was derived from."""
    seg = sc.selected.copy()
    corrected = np.arange(70, 90, dtype=np.int32)
    anchor = np.arange(5, 25, 3, dtype=np.int32)
    poses, corr, rows = orc.apply_explicit_correction(ctype, seg, sc.poses, corrected, anchor)
    assert rows.size == corrected.size * anchor.size
    assert np.array_equal(poses[:70], sc.poses[:70])
    # the first group moves rigidly by (R(theta), T): relative geometry inside the group is kept
    d0 = np.linalg.norm(sc.poses[71, :2] - sc.poses[88, :2])
    d1 = np.linalg.norm(poses[71, :2] - poses[88, :2])
    assert abs(d0 - d1) < 1e-4
    assert np.allclose(poses[70:, 2] - sc.poses[70:, 2], corr[2], atol=1e-6)
    # the tail moves rigidly too (about the ALREADY MOVED last corrected pose, plus the translation once more --
    # ApplyExplicitCorrection.cpp:404-413 as written -- so it keeps its own shape, not its offset to that pose)
    d0 = np.linalg.norm(sc.poses[100, :2] - sc.poses[92, :2])
    d1 = np.linalg.norm(poses[100, :2] - poses[92, :2])
    assert abs(d0 - d1) < 1e-4
    L = O.lib()
    tgt = np.zeros(4, np.float32)
    for r in rows[::7]:
        row = abi.HumanConstraint(int(r["type"]), int(r["constrained_pose_id"]), int(r["anchor_pose_id"]), float(r["delta_parallel"]),
                                  float(r["delta_perpendicular"]), float(r["delta_angle"]), float(r["relative_penalty_dir"]))
        ap = np.ascontiguousarray(poses[r["anchor_pose_id"]])
        L.orc_hc_target(C.byref(row), ap.ctypes.data_as(C.POINTER(C.c_float)), tgt.ctypes.data_as(C.POINTER(C.c_float)))
        cp = poses[r["constrained_pose_id"]]
        assert abs(tgt[0] - cp[0]) < 2e-5 and abs(tgt[1] - cp[1]) < 2e-5
        assert abs(np.angle(np.exp(1j * (tgt[2] - cp[2])))) < 2e-6
    if ctype == 2:     # line to line: the midpoint of selection A lands on the midpoint of selection B
        A0, A1, B0, B1 = seg.reshape(4, 2)
        th = corr[2]
        assert abs(abs(th) - np.arccos(np.clip(np.dot((A1 - A0) / np.linalg.norm(A1 - A0), (B1 - B0) / np.linalg.norm(B1 - B0)), -1, 1))) < 1e-6


def test_explicit_correction_rejects_point_and_corner(orc, sc):
    for t in (0, 1, 3):
        with pytest.raises(Exception):
            orc.apply_explicit_correction(t, sc.selected, sc.poses, [70, 71], [3])


def _counts(n, first, second):
    c = np.zeros((n, 2), np.int32)
    c[list(first), 0] = 9
    c[list(second), 1] = 9
    c[0, 0] = 5          # exactly 5 inliers is not enough (EMinput.cpp:309)
    return c


@pytest.mark.parametrize("which", ["oracle", "product"])
def test_order_user_input_cases(which):
    n = 60
    seg = np.arange(8, dtype=np.float32)
    if which == "oracle":
        o = O.OracleSolver()
        order = o.order_user_input
    else:
        from hitl_slam_amd import api
        lib = api.load_library()
        holder = type("H", (api.SolverBase,), dict(_cycle_names=api.Solver._cycle_names, _cycle_handle=True))
        h = holder.__new__(holder)
        h._lib = lib
        h._detail = lambda: ""
        order = h.order_user_input
    # good user: first selection seen late, second early
    r = order(_counts(n, range(40, 50), range(5, 20)), seg)
    assert list(r["corrected"]) == list(range(40, 50)) and list(r["anchor"]) == list(range(5, 20))
    assert r["bounds"] == (20, 39) and np.array_equal(r["seg_xy8"], seg)
    # bad user: selections swapped, and so are the points
    r = order(_counts(n, range(5, 20), range(40, 50)), seg)
    assert list(r["corrected"]) == list(range(40, 50)) and list(r["anchor"]) == list(range(5, 20))
    assert r["bounds"] == (20, 39) and np.array_equal(r["seg_xy8"], np.r_[seg[4:], seg[:4]])
    # partial overlap: the shared poses leave both lists
    r = order(_counts(n, range(30, 50), range(5, 34)), seg)
    assert list(r["corrected"]) == list(range(34, 50)) and list(r["anchor"]) == list(range(5, 30))
    assert r["bounds"] == (30, 33)
    # first inside second: only the second list shrinks, what is left straddles the first -> overlap error
    r = order(_counts(n, range(20, 25), range(10, 40)), seg)
    assert r["bounds"] == (-1, -1) and r["corrected"].size == 0
    # complete overlap
    r = order(_counts(n, range(20, 25), range(20, 25)), seg)
    assert r["bounds"] == (-1, -1)
    # a selection nobody sees
    with pytest.raises(Exception):
        order(_counts(n, [], range(20, 25)), seg)
    if which == "oracle":
        o.close()


def test_fit_segments_snaps_to_a_wall(orc, sc):
    """The fit rotates each selection about its midpoint onto the observations within 3 cm; DistanceToLineSegment's
    't > 1' is kept as written, so the inliers are the first metre from p0 plus a 3 cm disc at p1."""
    seg, inl, passes, its = orc.fit_segments(sc.poses, sc.selected.reshape(2, 4))
    assert (inl > 20).all() and (passes >= 1).all() and (passes <= 8).all()
    wx, wy = sc.world
    for k in range(2):
        p0, p1 = sc.selected[4 * k:4 * k + 2], sc.selected[4 * k + 2:4 * k + 4]
        q0, q1 = seg[k, :2], seg[k, 2:]
        # same midpoint and length
        assert np.allclose((p0 + p1) / 2, (q0 + q1) / 2, atol=1e-5)
        assert abs(np.linalg.norm(p1 - p0) - np.linalg.norm(q1 - q0)) < 1e-5
        # wall is along x: the fitted segment is within a few mrad of it
        d = (q1 - q0) / np.linalg.norm(q1 - q0)
        assert abs(d[1]) < 0.02
    # count check against a numpy restatement of the inlier rule on the final segment
    k = 0
    p0, p1 = seg[k, :2].astype(np.float32), seg[k, 2:].astype(np.float32)
    u = (p1 - p0) / np.float32(np.sqrt(np.float32((p1 - p0) @ (p1 - p0))))
    q = np.c_[wx, wy].astype(np.float32) - p0
    t = q @ u
    dist = np.where(t < 0, np.linalg.norm(q, axis=1), np.where(t > 1, np.linalg.norm(np.c_[wx, wy] - p1, axis=1), np.abs(q @ np.array([-u[1], u[0]], np.float32))))
    # inl[k] is the count of the LAST pass, taken at the segment before its final (sub-5 cm) adjustment
    assert abs(int((dist < 0.03).sum()) - int(inl[k])) <= max(6, inl[k] // 5)


def test_fit_segments_degenerate(orc, sc):
    with pytest.raises(Exception):
        orc.fit_segments(sc.poses, [[1.0, 2.0, 1.0, 2.0]])
    # a segment far from every observation: no inliers, theta stays acos(|dx| / len) -> a deterministic re-draw
    seg, inl, passes, its = orc.fit_segments(sc.poses, [[500.0, 500.0, 503.0, 496.0]])
    assert inl[0] == 0 and its[0] == 0
    assert np.allclose(seg[0], [503.0, 500.0, 500.0, 496.0], atol=1e-4)   # icm +- len * (|dx|, |dy|) / len


def test_full_cycle_on_the_oracle(orc, sc):
    r = orc.run_correction(4, sc.selected, sc.poses, sc.cov, [])
    assert r["status"] == "ok"
    lo, hi = r["order"]["bounds"]
    assert 0 <= lo < hi
    assert len(r["constraint_groups"]) == 1 and r["constraint_groups"][0].size == r["order"]["corrected"].size * r["order"]["anchor"].size
    assert r["summary"].termination == abi.CONVERGENCE
    # a colinear constraint between two pieces of one straight wall: the first-selected piece, as drawn by the corrected
    # poses, ends up on the line through the second selection
    o2 = O.OracleSolver()
    sc.prob.load(o2)
    off = np.asarray(sc.prob.off)
    pose_of = np.repeat(np.arange(sc.n), np.diff(off))
    lx, ly = sc.local
    on_piece = (np.abs(ly - 1.2) < 0.04) & (lx > 18.8) & (lx < 25.2) & np.isin(pose_of, r["order"]["corrected"])
    B0, B1 = r["order"]["seg_xy8"][4:6], r["order"]["seg_xy8"][6:8]
    nrm = np.array([-(B1 - B0)[1], (B1 - B0)[0]]) / np.linalg.norm(B1 - B0)

    def offset(poses):
        wx, wy = o2.transform_scans(poses)
        return np.abs((np.c_[wx, wy][on_piece] - B0) @ nrm).mean()
    before, after = offset(sc.poses), offset(r["poses"])
    o2.close()
    assert after < 0.02 and after < 0.6 * before, (before, after)
    # clicked points that are not on the map are refused before anything moves
    bad = sc.selected.copy()
    bad[0] += 3.0
    bad[1] += 40.0
    r2 = orc.run_correction(4, bad, sc.poses, sc.cov, [])
    assert r2["status"] == "not verified" and r2["poses"] is not None and len(r2["constraint_groups"]) == 0


def test_order_user_input_product_equals_oracle_on_random_counts():
    """The product's index logic (sorted-merge) against the oracle's literal nested loops / erase-remove on random
    observation counts, including empty selections, full overlap and interleaved sets."""
    from hitl_slam_amd import api
    lib = api.load_library()
    holder = type("H", (api.SolverBase,), dict(_cycle_names=api.Solver._cycle_names, _cycle_handle=True))
    h = holder.__new__(holder)
    h._lib = lib
    h._detail = lambda: ""
    o = O.OracleSolver()
    rng = np.random.default_rng(77)
    seg = np.arange(8, dtype=np.float32)
    agree = 0
    for trial in range(300):
        n = int(rng.integers(2, 80))
        kind = trial % 5
        counts = np.zeros((n, 2), np.int32)
        if kind == 0:      # random sprinkles
            counts[:, 0] = rng.integers(0, 12, n)
            counts[:, 1] = rng.integers(0, 12, n)
        elif kind == 1:    # two blocks, random order
            a, b = sorted(rng.integers(0, n, 2))
            counts[:a + 1, int(rng.integers(0, 2))] = 9
            counts[b:, int(rng.integers(0, 2))] = 9
        elif kind == 2:    # identical selections
            m = rng.random(n) < 0.4
            counts[m, 0] = 9
            counts[m, 1] = 9
        elif kind == 3:    # one selection inside the other
            counts[n // 4: 3 * n // 4 + 1, 0] = 9
            counts[n // 3: n // 2 + 1, 1] = 9
        else:              # one empty
            counts[rng.random(n) < 0.5, int(rng.integers(0, 2))] = 9
        res = []
        for order in (h.order_user_input, o.order_user_input):
            try:
                r = order(counts, seg)
                res.append((r["bounds"], list(r["corrected"]), list(r["anchor"]), r["seg_xy8"].tolist()))
            except api.HitlError as e:
                res.append(("error", e.status if hasattr(e, "status") else None))
        assert res[0][0] == res[1][0], (trial, kind, res)
        if res[0][0] != "error":
            assert res[0] == res[1], (trial, kind)
            agree += 1
    assert agree > 100
    o.close()


def test_endpoint_fit_reaches_the_minimum_scipy_finds(orc, sc):
    """Independent check of AutomaticEndpointAdjustment: the same EM loop in numpy with scipy.optimize.least_squares as
    the M step (same residuals, same start acos(|dx| / length), same inlier rule with the reference's 't > 1'), against
    the restated Ceres loop.  Ceres stops on its 1e-6 function tolerance, so the endpoints agree to a millimetre, and
    the number of EM passes is the same."""
    from scipy.optimize import least_squares
    wx, wy = sc.world
    W32 = np.c_[wx, wy].astype(np.float32)
    for k in range(2):
        p0 = sc.selected[4 * k:4 * k + 2].astype(np.float32)
        p1 = sc.selected[4 * k + 2:4 * k + 4].astype(np.float32)
        seg, inl, passes, its = orc.fit_segments(sc.poses, [np.r_[p0, p1]])
        a0, a1, n_pass = p0.copy(), p1.copy(), 0
        adj = (1.0, 1.0)
        while max(adj) > 0.05:
            n_pass += 1
            assert n_pass < 20
            d = a1 - a0
            u = d / np.float32(np.sqrt(np.float32(d @ d)))
            q = W32 - a0
            t = q @ u
            dist = np.where(t < 0, np.linalg.norm(q, axis=1),
                            np.where(t > 1, np.linalg.norm(W32 - a1, axis=1), np.abs(q @ np.array([-u[1], u[0]], np.float32))))
            pts = W32[dist.astype(np.float64) < 0.03].astype(np.float64)
            e0, e1 = a0.astype(np.float64), a1.astype(np.float64)
            cm, hy = (e0 + e1) / 2.0, float(np.linalg.norm(e0 - e1))
            length = hy / 2.0

            def resid(th):
                a = np.array([np.cos(th[0]), np.sin(th[0])])
                s1, s2 = cm + length * a, cm - length * a
                dd = s2 - s1
                tt = ((pts - s1) @ dd) / (dd @ dd)
                return np.linalg.norm(pts - (s1 + np.clip(tt, 0.0, 1.0)[:, None] * dd), axis=1)
            th0 = np.arccos(abs(e0[0] - e1[0]) / hy)
            th = least_squares(resid, [th0], method="trf", xtol=1e-12, ftol=1e-12, gtol=1e-12).x[0] if len(pts) else th0
            a = np.array([np.cos(th), np.sin(th)])
            n0, n1 = (cm + length * a).astype(np.float32), (cm - length * a).astype(np.float32)
            adj = (float(np.linalg.norm(a0 - n0)), float(np.linalg.norm(a1 - n1)))
            a0, a1 = n0, n1
        assert n_pass == int(passes[0])
        assert np.abs(np.r_[a0, a1] - seg[0]).max() < 1e-3, (np.r_[a0, a1], seg[0])
