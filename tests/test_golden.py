"""Committed golden fixtures (tests/golden/*.npz).  Since round 3 they are emitted by tests/golden/mirror.py, an independent
numpy restatement of the reference's path (literal autodiff functors, dense normal equations, the recursion of kdtree.cpp as
written) that shares no code with the C oracle or the product; the reference itself has no vectors (parity unpinned).
CPU: the C oracle reproduces them -- index sets bit for bit, fp64 to 1e-11, the whole Levenberg-Marquardt path iteration for
iteration, under both orders of the tolerance tests (hitl_options.tolerances_after_accept).  GPU: the HIP path through the C
ABI does the same within the north-star tolerance."""
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


def _check(solver, z, lin_rtol, matcher, prefix="solve", bf_matcher=None):
    if bf_matcher is not None:                             # the brute-force definition of the query (kd_tree_search = 0)
        bf_matcher(solver, z)
        got = solver.get_correspondences()
        for a, key in zip(got, ("pair_src", "pair_tgt", "match_offset", "src_idx", "tgt_idx")):
            assert np.array_equal(a, z["bf_" + key].astype(a.dtype)), "bf_" + key
    if matcher is not None:                                # the reference's KD-tree walk (the default)
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
        exp = z["%s_summary_%d" % (prefix, mode)]
        ec = z["%s_cost_%d" % (prefix, mode)]
        assert abs(summ.initial_cost - ec[0]) <= 1e-9 * ec[0]
        assert [summ.num_residuals, summ.num_hc_residuals] == exp.tolist()[3:]
        offband = int(np.abs(z["pair_src"].astype(np.int64) - z["pair_tgt"]).max()) > 4   # linear solve = PCG (1e-12), not the exact factorisation
        if mode == abi.MODE_STF_ONLY and (offband or _stf_components(z) > 1 or exp[0] == abi.NO_CONVERGENCE):
            # The one case that is NOT held to the step-for-step bar.  Mode B is the dormant STF-only solve
            # (JointOptimization.cpp:1156-1256: no odometry, only pose 0 held); on these corridor maps it is so ill-conditioned
            # that it never converges (100 iterations, NO_CONVERGENCE) and its accept/reject sequence depends on the ROUNDING of the
            # linear solver: the mirror itself takes 86 / 87 / 88 / 89 accepted steps of 100 with an eigen-decomposition / refined
            # Cholesky / LU / plain Cholesky solve of the same normal equations (final costs equal to 4 digits, poses 3e-4 m apart).
            # With more than one connected component of the scan-overlap graph J^T J is singular outright (a free rigid body per
            # component without pose 0: shown on the oracle's dense Hessian in tests/test_gpu_parity.py).  No implementation
            # can be held to another's path here; asserted: the cost level and a 1e-2 pose bar.  (Oracle and product do walk the
            # same path on connected maps with every pair inside the band -- banded Cholesky and cyclic reduction round alike --
            # see tests/test_gpu_parity.py.)
            assert abs(summ.final_cost - ec[1]) <= 2e-3 * ec[1]
            assert np.abs(poses - z["%s_poses_%d" % (prefix, mode)]).max() <= 1e-2
            continue
        assert [summ.termination, summ.iterations, summ.successful_steps] == exp.tolist()[:3]
        assert abs(summ.final_cost - ec[1]) <= 1e-7 * ec[1]
        assert np.abs(poses[:, :2] - z["%s_poses_%d" % (prefix, mode)][:, :2]).max() <= POSE_TOL_M
        assert np.abs(poses[:, 2] - z["%s_poses_%d" % (prefix, mode)][:, 2]).max() <= POSE_TOL_RAD


def test_fixtures_exist():
    assert len(GOLDEN) >= 3


def _options(late):
    opt = abi.default_options()
    opt.tolerances_after_accept = int(late)
    return opt


@pytest.mark.parametrize("late", [0, 1], ids=["tolerances_before_accept", "tolerances_after_accept"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_oracle_reproduces_golden(path, late):
    from oracle import hitl_oracle as O
    z = np.load(path)
    s = _load(O.OracleSolver(_options(late)), z, with_corr=False)
    _check(s, z, 1e-11, lambda sv, zz: sv.find_correspondences(zz["init"], int(zz["window"]), 1), "late_solve" if late else "solve",
           bf_matcher=lambda sv, zz: sv.find_correspondences(zz["init"], int(zz["window"]), 0))


@pytest.mark.gpu
@pytest.mark.parametrize("late", [0, 1], ids=["tolerances_before_accept", "tolerances_after_accept"])
@pytest.mark.parametrize("path", GOLDEN, ids=[os.path.basename(p)[:-4] for p in GOLDEN])
def test_hip_path_reproduces_golden(path, late):
    import hitl_slam_amd as H
    z = np.load(path)
    opt = _options(late)
    opt.kd_tree_search = 0
    s = _load(H.Solver(opt), z, with_corr=False)
    s.find_correspondences(z["init"], int(z["window"]))
    for a, key in zip(s.get_correspondences(), ("pair_src", "pair_tgt", "match_offset", "src_idx", "tgt_idx")):
        assert np.array_equal(a, z["bf_" + key].astype(a.dtype)), "bf_" + key
    s.close()
    s = _load(H.Solver(_options(late)), z, with_corr=False)
    _check(s, z, 1e-10, lambda sv, zz: sv.find_correspondences(zz["init"], int(zz["window"])), "late_solve" if late else "solve")
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
