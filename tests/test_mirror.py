"""The C oracle against tests/golden/mirror.py, the independent numpy restatement of the reference's path (literal functors
over forward-mode duals, dense normal equations, kdtree.cpp's recursion as written, Eigen's operation order for the relative
transform).  Two readings of the same reference text by different routes have to agree: functor values and Jacobians, the
fp32 factor set-up bit for bit, correspondence index sets bit for bit, and the whole Levenberg-Marquardt path.  (The
committed fixtures in tests/golden/ come from the mirror; tests/test_golden.py holds oracle and HIP path to those.)"""
import os
import sys

import numpy as np
import pytest

from conftest import Problem
from hitl_slam_amd import abi
from oracle import hitl_oracle as O

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import mirror as M  # noqa: E402


def test_the_mirror_is_independent_of_the_oracle_and_the_product():
    import ast
    tree = ast.parse(open(M.__file__).read())
    mods = set()
    for node in ast.walk(tree):
        if isinstance(node, ast.Import):
            mods |= {a.name.split(".")[0] for a in node.names}
        elif isinstance(node, ast.ImportFrom):
            mods.add((node.module or "").split(".")[0])
    assert mods <= {"ctypes", "math", "numpy"}, mods
    assert "libhitl" not in open(M.__file__).read()


def _hc_array(rows, n):
    return np.array([[r.type, r.constrained_pose_id, r.anchor_pose_id, r.delta_parallel, r.delta_perpendicular, r.delta_angle,
                      r.relative_penalty_dir] for r in rows[:n]], np.float64).reshape(n, 7)


def test_functors_agree(small_problem):
    rng = np.random.default_rng(3)
    for _ in range(40):
        prev = (rng.standard_normal(3) * [30, 30, 2]).astype(np.float32)
        cur = prev.copy()
        if rng.random() > 0.2:                             # the degenerate branch of the set-up (no translation) otherwise
            cur[:2] += (rng.standard_normal(2) * 0.3).astype(np.float32)
        cur[2] += np.float32(rng.standard_normal() * 0.2 + (6.0 if rng.random() < 0.2 else 0.0))
        A, rho, rot = M.odometry_setup(prev, cur)
        oA, orho, orot = O.odom_setup(prev, cur)
        assert np.array_equal(A.reshape(-1).view(np.uint32), oA.view(np.uint32))            # fp32 set-up: bit for bit
        assert np.float32(rho).view(np.uint32) == np.float32(orho).view(np.uint32)
        assert np.float32(rot).view(np.uint32) == np.float32(orot).view(np.uint32)
        x1 = prev.astype(np.float64) + rng.standard_normal(3) * 0.05
        x2 = cur.astype(np.float64) + rng.standard_normal(3) * 0.05
        r, J = M.pose_constraint(x1, x2, A, (0.03, 0.03, 0.01), rho, rot)
        orr, oJ = O.odom_eval(x1, x2, oA, orho, orot)
        assert np.abs(r - orr).max() <= 1e-11 * max(1.0, np.abs(orr).max()) and np.abs(J - oJ).max() <= 1e-9 * np.abs(oJ).max()
    prob = small_problem
    P = np.stack([prob.px, prob.py], 1)
    N = np.stack([prob.nx, prob.ny], 1)
    for _ in range(10):
        i, j = rng.integers(0, prob.n, 2)
        m = int(rng.integers(11, 60))
        a = prob.off[i] + rng.integers(0, 90, m)
        b = prob.off[j] + rng.integers(0, 90, m)
        x0, x1 = prob.gt[i] + rng.standard_normal(3) * 0.02, prob.gt[j] + rng.standard_normal(3) * 0.02
        r, J = M.point_to_point_glob(x0, x1, P[a], P[b], N[a], N[b], 0.05, 1.0 / 40.0)
        orr, oJ = O.stf_eval(x0, x1, P[a], P[b], N[a], N[b])
        assert np.abs(r - orr).max() <= 1e-12 * np.abs(orr).max() and np.abs(J - oJ).max() <= 1e-10 * np.abs(oJ).max()
    rows, nrows = Problem(20, 16, ppf=20, groups=2, kind=0).rows
    poses = Problem(20, 16, ppf=20, groups=2, kind=0).init
    for k in range(nrows):
        row = _hc_array(rows, nrows)[k]
        typ, tx, ty, tt, pen = M.human_target(row, poses)
        ot = O.hc_target(rows[k], poses[int(row[2])])
        assert np.array_equal(np.array([tx, ty, tt, pen], np.float32).view(np.uint32), ot.view(np.uint32))      # float targets: bit for bit
        x = poses[int(row[1])].astype(np.float64) + rng.standard_normal(3) * 0.1
        r, J = M.human_row(typ, (tx, ty, tt, pen), x)
        orr, oJ = O.hc_eval(typ, ot, x)
        assert r.shape == orr.shape and np.abs(r - orr).max() <= 1e-13 and np.abs(J - oJ).max() <= 1e-13


@pytest.mark.parametrize("case", [(14, 40, 14, 3, 1, 1, 20260105), (10, 36, 5, 4, 1, 1, 20260106), (9, 30, 9, 0, 2, 1, 20260107),
                                   (12, 33, 12, 2, 1, 0, 20260108)])
def test_correspondence_sets_agree(case):
    """Both search methods; window 4 has 8 targets per source against a cap of 6 (the per-point cap binds), window 0 is the
    reference's all-pairs loop, skip 2 leaves every other source point out; poses far from the origin (the relative transform
    subtracts two large translations in the reference's order)."""
    n, m, ppf, window, skip, far, seed = case
    prob = Problem(n, m, ppf=ppf, groups=1, seed=seed)
    if far:
        prob.init[:, 0] += np.float32(731.25)
        prob.init[:, 1] -= np.float32(1290.5)
    opt = abi.default_options()
    opt.num_skip_readings = skip
    o = prob.load(O.OracleSolver(opt))
    for method in (1, 0):
        o.find_correspondences(prob.init, window, method)
        got = M.find_stf_correspondences(prob.init, prob.off, prob.px, prob.py, prob.nx, prob.ny, window, method, skip=skip)
        for a, b in zip(got, o.get_correspondences()):
            assert np.array_equal(a, b)
        assert len(got[0]) > 0
    o.close()


@pytest.mark.parametrize("late", [0, 1])
@pytest.mark.parametrize("case", [(14, 40, 14, 1, 0, 20260105), (18, 24, 9, 2, 4, 20260111), (8, 30, 8, 1, 2, 20260112)])
def test_solves_agree(case, late):
    n, m, ppf, groups, kind, seed = case
    prob = Problem(n, m, ppf=ppf, groups=groups, kind=kind, seed=seed)
    opt = abi.default_options()
    opt.tolerances_after_accept = late
    o = prob.load(O.OracleSolver(opt))
    o.find_correspondences(prob.init, 3, 1)
    corr = o.get_correspondences()
    hc = _hc_array(*prob.rows)
    rng = np.random.default_rng(seed)
    xlin = prob.init.astype(np.float64) + 0.01 * rng.standard_normal(prob.init.shape)
    for mode in (0, 1, 2):
        P = M.Problem(mode, prob.init, hc, corr, (prob.off, prob.px, prob.py, prob.nx, prob.ny))
        cost, g, pr, pj = M.linearize(P, xlin)
        oc, og, opr, opj = o.linearize(mode, prob.init, xlin)
        assert abs(cost - oc) <= 1e-12 * oc and np.abs(g - og).max() <= 1e-11 * np.abs(og).max()
        if mode:
            assert np.abs(pr - opr).max() <= 1e-12 * np.abs(opr).max() and np.abs(pj - opj).max() <= 1e-10 * np.abs(opj).max()
        s = M.solve(P, tolerances_after_accept=bool(late))
        po, so = o.solve(mode, prob.init)
        assert abs(s["initial_cost"] - so.initial_cost) <= 1e-12 * so.initial_cost
        assert (s["num_residuals"], s["num_hc_residuals"]) == (so.num_residuals, so.num_hc_residuals)
        if mode == abi.MODE_STF_ONLY and so.termination == abi.NO_CONVERGENCE:
            # the dormant STF-only solve does not converge and its path depends on the rounding of the linear solver
            # (tests/test_golden.py): cost level only
            assert abs(s["final_cost"] - so.final_cost) <= 5e-3 * so.final_cost
            continue
        assert (s["termination"], s["iterations"], s["successful_steps"]) == (so.termination, so.iterations, so.successful_steps)
        assert abs(s["final_cost"] - so.final_cost) <= 1e-9 * so.final_cost
        assert np.abs(s["poses"] - po).max() <= 1e-6
    o.close()


# ---------------------------------------------------------------------------------------------------------------------
# round 4: the correction cycle and the covariance recovery (tests/golden/mirror_cycle.py) against the oracle
# ---------------------------------------------------------------------------------------------------------------------
import mirror_cycle as MC  # noqa: E402


def test_the_cycle_mirror_and_the_fixture_maker_are_independent_of_the_oracle():
    import ast
    here = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
    for name, allowed in (("mirror_cycle.py", {"ctypes", "math", "numpy", "mirror"}),
                          ("make_golden_cycle.py", {"os", "sys", "numpy", "mirror", "mirror_cycle", "hitl_slam_amd", "tools"}),
                          ("make_golden.py", None)):
        src = open(os.path.join(here, name)).read()
        mods = set()
        for node in ast.walk(ast.parse(src)):
            if isinstance(node, ast.Import):
                mods |= {a.name.split(".")[0] for a in node.names}
            elif isinstance(node, ast.ImportFrom):
                mods.add((node.module or "").split(".")[0])
        assert "oracle" not in mods, (name, mods)
        if allowed is not None:
            assert mods <= allowed, (name, mods)


def _eq32(a, b):
    return np.array_equal(np.asarray(a, np.float32).view(np.uint32), np.asarray(b, np.float32).view(np.uint32))


@pytest.mark.parametrize("ctype", [2, 4, 5, 6])
@pytest.mark.parametrize("swap", [False, True])
def test_correction_cycle_stage_by_stage_against_the_oracle(ctype, swap):
    """One user correction through every stage of HitLSLAM::Run, mirror (literal float32 loops) against oracle (closed forms,
    explicit index arithmetic), all four correction types, and with the two selections given in the other order (the 'bad
    user' branch of OrderAndFilterUserInput, which swaps them back): index sets, fp32 poses, covariances and rows BIT FOR BIT."""
    from tests.cycle_scenario import Scenario
    sc = Scenario(n_poses=70, obs=60, seed=20260107 + ctype, first_piece=(10.5, 17.5))
    pr = sc.prob
    clicks = sc.selected.copy()
    if swap:
        clicks = np.concatenate([clicks[4:], clicks[:4]])
    o = O.OracleSolver()
    pr.load(o)
    wx, wy = MC.transform_clouds(sc.poses, pr.off, pr.px, pr.py)
    owx, owy = o.transform_scans(sc.poses)
    assert _eq32(wx, owx) and _eq32(wy, owy)
    assert np.array_equal(MC.verify_points(wx, wy, clicks.reshape(-1, 2), 0.05), o.verify_points(sc.poses, clicks.reshape(-1, 2), 0.05))
    seg, inl, pas, its = MC.fit_segments(wx, wy, clicks.reshape(2, 4))
    oseg, oinl, opas, oits = o.fit_segments(sc.poses, clicks.reshape(2, 4))
    assert _eq32(seg, oseg) and np.array_equal(inl, oinl) and np.array_equal(pas, opas) and np.array_equal(its, oits)
    counts = MC.observation_sets(wx, wy, pr.off, seg.reshape(-1), 0.03)
    ocounts, _ = o.observation_sets(sc.poses, seg.reshape(-1), 0.03, want_mask=False)
    assert np.array_equal(counts, ocounts)
    od, ood = MC.order_user_input(counts, seg.reshape(-1)), o.order_user_input(ocounts, seg.reshape(-1))
    assert np.array_equal(od["corrected"], ood["corrected"]) and np.array_equal(od["anchor"], ood["anchor"])
    assert od["bounds"] == tuple(ood["bounds"]) and _eq32(od["seg_xy8"], ood["seg_xy8"]) and od["bounds"][0] >= 0
    if swap:
        assert _eq32(od["seg_xy8"][:4], seg.reshape(-1)[4:])        # the selections were swapped back
    p2, corr, rows = MC.apply_explicit_correction(ctype, od["seg_xy8"], sc.poses, od["corrected"], od["anchor"])
    op2, ocorr, orows = o.apply_explicit_correction(ctype, ood["seg_xy8"], sc.poses, ood["corrected"], ood["anchor"])
    assert _eq32(p2, op2) and _eq32(corr, ocorr) and len(rows) == len(orows)
    for k, f in enumerate(("type", "constrained_pose_id", "anchor_pose_id")):
        assert np.array_equal(np.array([r[k] for r in rows]), orows[f])
    for k, f in ((3, "delta_parallel"), (4, "delta_perpendicular"), (5, "delta_angle"), (6, "relative_penalty_dir")):
        assert _eq32(np.array([r[k] for r in rows], np.float32), orows[f]), f
    p3, cov3 = MC.backprop(p2, sc.cov, od["bounds"][0], od["bounds"][1], corr)
    op3, ocov3 = o.backprop(op2, sc.cov, ood["bounds"][0], ood["bounds"][1], ocorr)
    assert _eq32(p3, op3) and _eq32(cov3, ocov3)
    assert _eq32(MC.normalize_angles(p3), o.normalize_angles(op3))
    o.close()


def test_covariance_blocks_against_the_oracle():
    """vector_mapping.cpp:2772-2812: dense pseudo-inverse of the mirror's J^T J against the oracle's banded inverse, robot frame."""
    prob = Problem(40, 30, ppf=20, groups=2, kind=0)
    o = O.OracleSolver()
    prob.load(o)
    po, _ = o.solve(abi.MODE_ODOM_HC, prob.init)
    So = o.pose_covariances(abi.MODE_ODOM_HC, po)
    rows = prob.rows[0][:prob.rows[1]]
    hc = [(int(r.type), int(r.constrained_pose_id), int(r.anchor_pose_id), np.float32(r.delta_parallel), np.float32(r.delta_perpendicular),
           np.float32(r.delta_angle), np.float32(r.relative_penalty_dir)) for r in rows]
    Sm = MC.pose_covariances(M.Problem(M.MODE_ODOM_HC, po, hc, None, None), po)
    scale = np.abs(So).max(axis=(1, 2))[:, None, None] + 1e-300
    assert (np.abs(Sm - So) / scale).max() < 1e-7
    o.close()
