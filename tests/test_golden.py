"""Committed golden fixtures (tests/golden/*.npz, made by tests/golden/make_golden.py from
the oracle; the reference itself has none -- parity unpinned).  CPU: the oracle still
reproduces them bit-for-bit in its index work and to 1e-12 in fp64.  GPU: the HIP path
through the C ABI lands on them within the north-star tolerance."""
import glob
import os

import numpy as np
import pytest

from hitl_slam_amd import abi
from hitl_slam_amd.api import hc_rows

_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")
GOLDEN = sorted(p for p in glob.glob(os.path.join(_DIR, "*.npz")) if not os.path.basename(p).startswith("cycle_"))
CYCLE = os.path.join(_DIR, "cycle_corridor.npz")     # made by tests/golden/make_golden_cycle.py
POSE_TOL_M = 1e-4
POSE_TOL_RAD = 1e-4


def _load(solver, z, with_corr=True):
    solver.set_map(z["obs_offset"], z["px"], z["py"], z["nx"], z["ny"], z["cov9"])
    solver.set_human_constraints(hc_rows(z["hc"]))
    if with_corr:
        solver.set_correspondences(z["pair_src"], z["pair_tgt"], z["match_offset"], z["src_idx"].astype(np.int32),
                                   z["tgt_idx"].astype(np.int32))
    return solver


def _stf_components(z):
    """Connected components of the scan-overlap graph (poses joined by a correspondence pair)."""
    n = len(z["init"])
    parent = list(range(n))

    def find(a):
        while parent[a] != a:
            parent[a] = parent[parent[a]]
            a = parent[a]
        return a
    for a, b in zip(z["pair_src"], z["pair_tgt"]):
        parent[find(int(a))] = find(int(b))
    return len({find(i) for i in range(n)})


def _check(solver, z, lin_rtol, matcher):
    if matcher is not None:
        matcher(solver, z)
        got = solver.get_correspondences()
        for a, key in zip(got, ("pair_src", "pair_tgt", "match_offset", "src_idx", "tgt_idx")):
            assert np.array_equal(a, z[key].astype(a.dtype)), key
    for mode in (0, 1, 2):
        cost, g, pr, pj = solver.linearize(mode, z["init"], z["xlin"])
        assert abs(cost - z["lin_cost_%d" % mode]) <= lin_rtol * abs(z["lin_cost_%d" % mode])
        assert np.abs(g - z["lin_grad_%d" % mode]).max() <= lin_rtol * np.abs(z["lin_grad_%d" % mode]).max()
        if mode:
            assert np.abs(pr - z["lin_pair_r_%d" % mode]).max() <= lin_rtol * np.abs(z["lin_pair_r_%d" % mode]).max()
            assert np.abs(pj - z["lin_pair_J_%d" % mode]).max() <= lin_rtol * np.abs(z["lin_pair_J_%d" % mode]).max()
        poses, summ = solver.solve(mode, z["init"])
        exp = z["solve_summary_%d" % mode]
        ec = z["solve_cost_%d" % mode]
        assert abs(summ.initial_cost - ec[0]) <= 1e-9 * ec[0]
        assert [summ.num_residuals, summ.num_hc_residuals] == exp.tolist()[3:]
        offband = int(np.abs(z["pair_src"].astype(np.int64) - z["pair_tgt"]).max()) > 4   # linear solve = PCG (1e-12), not the exact factorisation
        if mode == abi.MODE_STF_ONLY and (offband or _stf_components(z) > 1):
            # The one case that is NOT held to the step-for-step bar.  Mode B is the dormant STF-only solve
            # (JointOptimization.cpp:1156-1256: no odometry, only pose 0 held).  When the scan-overlap graph of the fixture has
            # more than one connected component (two floors that share no overlap), every component that does not hold pose 0
            # is a free rigid body: J^T J has a 3-dimensional null space per such component (shown on the oracle's dense
            # Hessian in tests/test_gpu_parity.py).  The LM diagonal makes the linear systems solvable but leaves the step
            # along those directions to rounding, so two exact solvers (banded Cholesky there, cyclic reduction / PCG here)
            # drift apart over ~100 accept/reject decisions; the same holds when pose pairs outside the band send the linear
            # solve through the PCG (relative residual 1e-12 instead of an exact factorisation) on this slowly converging
            # problem.  Asserted: the cost level and a 1e-2 pose bar.  On connected maps with every pair inside the band mode B
            # reproduces the oracle's whole run, iteration for iteration (tests/test_gpu_parity.py).
            assert abs(summ.final_cost - ec[1]) <= 2e-3 * ec[1]
            assert np.abs(poses - z["solve_poses_%d" % mode]).max() <= 1e-2
            continue
        assert [summ.termination, summ.iterations, summ.successful_steps] == exp.tolist()[:3]
        assert abs(summ.final_cost - ec[1]) <= 1e-7 * ec[1]
        assert np.abs(poses[:, :2] - z["solve_poses_%d" % mode][:, :2]).max() <= POSE_TOL_M
        assert np.abs(poses[:, 2] - z["solve_poses_%d" % mode][:, 2]).max() <= POSE_TOL_RAD


def test_fixtures_exist():
    assert len(GOLDEN) >= 3


@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path):
    from oracle import hitl_oracle as O
    z = np.load(path)
    s = _load(O.OracleSolver(), z, with_corr=False)
    _check(s, z, 1e-12, lambda sv, zz: sv.find_correspondences(zz["init"], int(zz["window"]), 0))


@pytest.mark.gpu
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_reproduces_golden(path):
    import hitl_slam_amd as H
    z = np.load(path)
    s = _load(H.Solver(), z, with_corr=False)
    _check(s, z, 1e-10, lambda sv, zz: sv.find_correspondences(zz["init"], int(zz["window"])))
    s.close()


def _cycle_check(s, z, exact_libm):
    """One whole user correction (HitLSLAM::Run) stage by stage against the fixture."""
    s.set_map(z["obs_offset"], z["px"], z["py"], z["nx"], z["ny"], z["cov9"])
    eq = lambda a, b: np.array_equal(np.asarray(a).view(np.uint32), np.asarray(b).view(np.uint32))
    assert np.array_equal(s.verify_points(z["init"], z["clicks"].reshape(-1, 2), 0.05), z["seen"])
    seg, inl, pas, its = s.fit_segments(z["init"], z["clicks"].reshape(2, 4))
    assert np.array_equal(inl, z["fit_inliers"]) and np.array_equal(pas, z["fit_passes"]) and np.array_equal(its, z["fit_iterations"])
    assert np.abs(seg - z["fitted"]).max() <= 2e-6
    counts, _ = s.observation_sets(z["init"], z["fitted"].reshape(-1), 0.03, want_mask=False)
    assert np.array_equal(counts, z["counts"])
    od = s.order_user_input(counts, z["fitted"].reshape(-1))
    assert np.array_equal(od["corrected"], z["corrected"]) and np.array_equal(od["anchor"], z["anchor"])
    assert tuple(z["bounds"]) == od["bounds"] and eq(od["seg_xy8"], z["seg_ordered"])
    p2, corr, rows = s.apply_explicit_correction(int(z["ctype"]), z["seg_ordered"], z["init"], z["corrected"], z["anchor"])
    assert eq(p2, z["poses_explicit"]) and eq(corr, z["correction"])
    for f in ("type", "constrained_pose_id", "anchor_pose_id"):
        assert np.array_equal(rows[f], z["rows"][f])
    for f in ("delta_parallel", "delta_perpendicular", "relative_penalty_dir"):
        assert eq(rows[f], z["rows"][f])
    if exact_libm:
        assert eq(rows["delta_angle"], z["rows"]["delta_angle"])
    else:     # three device libm calls deep
        assert (np.abs(rows["delta_angle"].astype(np.float64) - z["rows"]["delta_angle"]) <= 4 * np.spacing(np.abs(z["rows"]["delta_angle"])) + 1e-12).all()
    p3, cov3 = s.backprop(z["poses_explicit"], z["cov9"], int(z["bounds"][0]), int(z["bounds"][1]), z["correction"])
    assert eq(p3, z["poses_backprop"]) and eq(cov3, z["cov_backprop"])
    assert eq(s.normalize_angles(z["poses_backprop"]), z["poses_normalized"])
    s.set_human_constraints(z["rows"])
    p5, summ = s.solve(abi.MODE_ODOM_HC, z["poses_normalized"])
    assert summ.termination == int(z["termination"]) and summ.iterations == int(z["iterations"])
    assert np.abs(p5[:, :2] - z["poses_solved"][:, :2]).max() <= POSE_TOL_M
    assert np.abs(p5[:, 2] - z["poses_solved"][:, 2]).max() <= POSE_TOL_RAD
    S = s.pose_covariances(abi.MODE_ODOM_HC, z["poses_solved"])
    scale = np.abs(z["pose_covariances"]).max(axis=(1, 2))[:, None, None] + 1e-300
    assert (np.abs(S - z["pose_covariances"]) / scale).max() < 1e-6


def test_oracle_reproduces_cycle_golden():
    from oracle import hitl_oracle as O
    s = O.OracleSolver()
    _cycle_check(s, np.load(CYCLE), exact_libm=True)
    s.close()


@pytest.mark.gpu
def test_hip_path_reproduces_cycle_golden():
    import hitl_slam_amd as H
    s = H.Solver()
    _cycle_check(s, np.load(CYCLE), exact_libm=False)
    s.close()
