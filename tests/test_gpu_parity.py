"""GPU parity: the HIP path through the C ABI (hitl_slam_amd.Solver -> libhitlgpu.so) against
the CPU oracle on identical seeded inputs.  Bars: index sets bit-exact; residuals/Jacobians
1e-10 relative; converged poses 1e-4 m / 1e-4 rad (BASELINE.json north_star)."""
import numpy as np
import pytest

from conftest import Problem, oracle_corr

pytestmark = pytest.mark.gpu

POSE_TOL_M = 1e-4     # north_star tolerance, metres
POSE_TOL_RAD = 1e-4   # north_star tolerance, radians


def _gpu():
    import hitl_slam_amd as H
    return H.Solver()


def _oracle():
    from oracle import hitl_oracle as O
    return O.OracleSolver()


def _relerr(a, b):
    return np.abs(a - b).max() / max(np.abs(b).max(), 1e-300)


def test_extension_is_loaded_and_runs_on_device():
    import hitl_slam_amd as H
    lib = H.load_library()
    assert lib.hitl_abi_version() == H.abi.ABI_VERSION
    s = _gpu()
    s.close()


@pytest.mark.parametrize("mode", [0, 1, 2])
def test_linearize_matches_oracle(small_problem, mode):
    prob = small_problem
    corr = oracle_corr(prob)
    g, o = prob.load(_gpu(), corr=corr), prob.load(_oracle(), corr=corr)
    rng = np.random.default_rng(5)
    x = prob.init.astype(np.float64) + 0.01 * rng.standard_normal(prob.init.shape)
    cg, gg, rg, jg = g.linearize(mode, prob.init, x)
    co, go, ro, jo = o.linearize(mode, prob.init, x)
    assert abs(cg - co) <= 1e-11 * abs(co)
    assert _relerr(gg, go) <= 1e-10
    assert gg[:3].tolist() == [0.0, 0.0, 0.0]          # pose 0 is constant
    if mode != 0:
        assert _relerr(rg, ro) <= 1e-11
        assert _relerr(jg, jo) <= 1e-10


def test_matcher_bit_exact_vs_bruteforce_definition(small_problem):
    prob = small_problem
    g, o = prob.load(_gpu()), prob.load(_oracle())
    for window in (1, 3, 0):                              # 0 = all pairs (the reference's search)
        g.find_correspondences(prob.init, window)
        o.find_correspondences(prob.init, window, 0)
        for a, b in zip(g.get_correspondences(), o.get_correspondences()):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("mode", [0, 2])
def test_solve_matches_oracle(small_problem, mode):
    prob = small_problem
    corr = oracle_corr(prob)
    g, o = prob.load(_gpu(), corr=corr), prob.load(_oracle(), corr=corr)
    pg, sg = g.solve(mode, prob.init)
    po, so = o.solve(mode, prob.init)
    assert sg.termination == so.termination
    assert sg.iterations == so.iterations and sg.successful_steps == so.successful_steps
    assert sg.num_residuals == so.num_residuals and sg.num_hc_residuals == so.num_hc_residuals
    assert abs(sg.final_cost - so.final_cost) <= 1e-9 * max(so.final_cost, 1e-12)
    assert np.abs(pg[:, :2] - po[:, :2]).max() <= POSE_TOL_M
    assert np.abs(pg[:, 2] - po[:, 2]).max() <= POSE_TOL_RAD
    assert np.array_equal(pg[0], prob.init[0])            # pose 0 held constant


def test_stf_only_short_horizon_lockstep(small_problem):
    """Mode B (the reference's dormant PostHumanOptimization): chaotic over 100 iterations, so
    lock-step parity is checked over the first 12 LM iterations, where both implementations
    must take the same accept/reject decisions and land on the same poses."""
    import hitl_slam_amd as H
    prob = small_problem
    corr = oracle_corr(prob)
    opt = H.default_options()
    opt.max_num_iterations = 12
    from oracle import hitl_oracle as O
    g, o = prob.load(H.Solver(opt), corr=corr), prob.load(O.OracleSolver(opt), corr=corr)
    pg, sg = g.solve(H.MODE_STF_ONLY, prob.init)
    po, so = o.solve(H.MODE_STF_ONLY, prob.init)
    assert sg.termination == so.termination == H.NO_CONVERGENCE
    assert (sg.iterations, sg.successful_steps) == (so.iterations, so.successful_steps) == (12, so.successful_steps)
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.abs(pg - po).max() <= 1e-5


def test_stf_only_full_run_reaches_the_same_cost_level(small_problem):
    import hitl_slam_amd as H
    prob = small_problem
    corr = oracle_corr(prob)
    g, o = prob.load(_gpu(), corr=corr), prob.load(_oracle(), corr=corr)
    pg, sg = g.solve(H.MODE_STF_ONLY, prob.init)
    po, so = o.solve(H.MODE_STF_ONLY, prob.init)
    assert abs(sg.final_cost - so.final_cost) <= 2e-3 * so.final_cost
    assert np.abs(pg - po).max() <= 1e-2


def test_solve_with_offband_pairs_uses_pcg(small_problem):
    """All-pairs correspondences (the reference's search) leave pose pairs outside the band:
    the PCG must iterate and still land on the oracle's exact-Cholesky step sequence."""
    prob = Problem(24, 64, ppf=24, groups=1)
    corr = oracle_corr(prob, window=0)
    assert np.abs(corr[0] - corr[1]).max() > 4
    g, o = prob.load(_gpu(), corr=corr), prob.load(_oracle(), corr=corr)
    pg, sg = g.solve(2, prob.init)
    po, so = o.solve(2, prob.init)
    assert sg.num_offband_pairs > 0 and sg.linear_iterations > sg.iterations
    assert sg.termination == so.termination and sg.iterations == so.iterations
    assert np.abs(pg[:, :2] - po[:, :2]).max() <= POSE_TOL_M
    assert np.abs(pg[:, 2] - po[:, 2]).max() <= POSE_TOL_RAD


def test_no_constraint_start_is_a_fixed_point():
    """BASELINE config 1: no human constraints; the odometry measurements are re-derived from
    the current poses (JointOptimization.cpp:748-769), so the start is the optimum up to the
    fp32 rounding of those measurements (|r| ~ 1e-6): a few no-op LM iterations, poses
    unchanged at fp32 resolution, identical behaviour in oracle and HIP path."""
    import hitl_slam_amd as H
    prob = Problem(40, 32, ppf=20, groups=0)
    g, o = prob.load(_gpu()), prob.load(_oracle())
    pg, sg = g.solve(H.MODE_ODOM_HC, prob.init)
    po, so = o.solve(H.MODE_ODOM_HC, prob.init)
    assert sg.termination == H.CONVERGENCE == so.termination
    assert sg.iterations == so.iterations
    assert sg.initial_cost <= 1e-8 and so.initial_cost <= 1e-8
    assert np.abs(pg - prob.init).max() <= 2e-6
    assert np.abs(pg - po).max() <= 1e-6


def test_evaluate_gradient_matches_oracle(small_problem):
    prob = small_problem
    corr = oracle_corr(prob)
    g, o = prob.load(_gpu(), corr=corr), prob.load(_oracle(), corr=corr)
    cg, gg = g.evaluate(2, prob.init)
    co, go = o.evaluate(2, prob.init)
    assert abs(cg - co) <= 1e-11 * abs(co) and _relerr(gg, go) <= 1e-10


def test_ragged_and_empty_scans():
    """Ragged obs_offset, one empty scan, a pose with no pairs at all."""
    import hitl_slam_amd as H
    prob = Problem(30, 48, ppf=30, groups=1)
    keep = np.ones(len(prob.px), bool)
    M = 48
    keep[5 * M:6 * M] = False                  # pose 5: empty scan
    keep[7 * M + 10:8 * M] = False             # pose 7: 10 points
    counts = np.array([keep[i * M:(i + 1) * M].sum() for i in range(prob.n)])
    prob.off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    prob.px, prob.py, prob.nx, prob.ny = (a[keep] for a in (prob.px, prob.py, prob.nx, prob.ny))
    g, o = prob.load(_gpu()), prob.load(_oracle())
    g.find_correspondences(prob.init, 3)
    o.find_correspondences(prob.init, 3, 0)
    cg, co = g.get_correspondences(), o.get_correspondences()
    for a, b in zip(cg, co):
        assert np.array_equal(a, b)
    assert 5 not in set(cg[0].tolist()) | set(cg[1].tolist())
    pg, sg = g.solve(H.MODE_JOINT, prob.init)
    po, so = o.solve(H.MODE_JOINT, prob.init)
    assert sg.iterations == so.iterations
    assert np.abs(pg - po).max() <= POSE_TOL_M


def test_errors_are_loud():
    import hitl_slam_amd as H
    s = _gpu()
    with pytest.raises(H.HitlError):
        s.solve(0, np.zeros((0, 3), np.float32))          # no map yet
    prob = Problem(10, 16, ppf=10, groups=0)
    prob.load(s)
    with pytest.raises(H.HitlError):
        s.solve(H.MODE_STF_ONLY, prob.init)                # no correspondences
    with pytest.raises(H.HitlError):
        s.set_correspondences([0], [0], [0, 1], [0], [0])  # src == tgt
    s.close()


def test_in_library_rccl_allgather_world1():
    """Bring-up check of the built-in RCCL exchange (ncclGetUniqueId, ncclCommInitRank,
    ncclAllGather on the solver's stream) on the single GPU of the test box.  Runs in its own process, like every rank
    of a real job does: a process that has initialised RCCL should end when its work does, not live on inside a
    long test session that keeps creating and destroying other solvers."""
    import os
    import subprocess
    import sys
    code = (
        "import ctypes as C\n"
        "import hitl_slam_amd as H\n"
        "from hitl_slam_amd.api import comm_unique_id\n"
        "uid = comm_unique_id()\n"
        "assert len(uid) == 128 and any(uid)\n"
        "buf = (C.c_char * 128).from_buffer_copy(uid)\n"
        "opt = H.default_options()\n"
        "opt.rccl_unique_id = C.cast(buf, C.c_void_p)\n"
        "s = H.Solver(opt)\n"
        "s.comm_selftest()\n"
        "s.close()\n"
        "print('rccl self-test ok')\n")
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, HITL_FORCE_RCCL="1", PYTHONPATH=root + os.pathsep + os.environ.get("PYTHONPATH", ""))
    r = subprocess.run([sys.executable, "-c", code], capture_output=True, text=True, env=env, cwd=root, timeout=300)
    assert r.returncode == 0 and "rccl self-test ok" in r.stdout, (r.returncode, r.stdout[-400:], r.stderr[-800:])


def test_timing_levels_do_not_change_results(small_problem):
    """hitl_options.timing_level only decides how many device-timing events a solve records (0: none, 1: every 4th
    linearisation, 2: all + linear-solver time): poses, cost and iteration count are identical, the timers behave as
    documented."""
    import hitl_slam_amd as H
    prob = small_problem
    corr = oracle_corr(prob)
    res = []
    for level in (0, 1, 2):
        o = H.default_options()
        o.timing_level = level
        s = H.Solver(o)
        prob.load(s, corr=corr)
        p, summ = s.solve(H.MODE_JOINT, prob.init)
        res.append((p, summ.final_cost, summ.iterations, summ.linearize_timed_calls, summ.linearize_calls, summ.linear_solver_ms, summ.linearize_ms))
        s.close()
    for r in res[1:]:
        assert np.array_equal(r[0], res[0][0]) and r[1] == res[0][1] and r[2] == res[0][2]
    assert res[0][3] == 0 and res[0][5] == 0.0 and res[0][6] == 0.0
    assert 0 < res[1][3] <= (res[1][4] + 3) // 4 + 1 and res[1][5] == 0.0 and res[1][6] > 0.0
    assert res[2][3] == res[2][4] and res[2][5] > 0.0
