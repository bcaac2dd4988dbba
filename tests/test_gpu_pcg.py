"""The sparse PCG solve (systems with pose pairs outside the band: the reference's all-pairs search on a map that revisits places,
JointOptimization.cpp:561-642; solved by SPARSE_SCHUR at :145-162, 1208) -- since round 6 ONE persistent launch per linear solve
(hitl_slam_amd/csrc/hitl_pcg.hip).  Held to the oracle's exact solve (a banded Cholesky over the whole pair bandwidth) on a Figure-8
drive at a size the oracle finishes in seconds, to the launch-per-stage loop of rounds 2-5 (test hook pcg_fused = 0: the same LM
path), over every band width and over chain lengths on both sides of the fused launch's limits, and with the failure path forced
(pcg_starve: the launch runs without its top workgroup, the chunk workgroups give up, the host repeats that solve the old way)."""
import numpy as np
import pytest

import figure8
from conftest import Problem
from hitl_slam_amd import abi
from hitl_slam_amd.api import hc_rows

pytestmark = pytest.mark.gpu


def _pdiff(a, b):
    d = np.asarray(a, np.float64) - np.asarray(b, np.float64)
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    return float(np.abs(d[:, :2]).max()), float(np.abs(d[:, 2]).max())


def _fig8(n, m, laps):
    d = figure8.make(n, m, seed=7, laps=laps)
    rows = hc_rows(figure8.constraints(d["gt"], 5, 3))
    return d, rows


def _load(solver, d, rows):
    solver.set_map(d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
    solver.set_human_constraints(rows)
    return solver


def test_figure8_all_pairs_mode_c_equals_the_oracles_exact_solve():
    """BASELINE configs[2] at a size the oracle's dense factorisation finishes in seconds (400 poses x 120 points, two laps): the
    same correspondence sets, the same LM path, poses within 1e-4 m / 1e-4 rad -- and the linear solves really ran as the one
    persistent launch (hitl_summary.pcg_timed_calls)."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d, rows = _fig8(400, 120, 2)
    opt = H.default_options()
    opt.timing_level = 2
    g, o = _load(H.Solver(opt), d, rows), _load(O.OracleSolver(), d, rows)
    g.find_correspondences(d["init"], 0)
    o.find_correspondences(d["init"], 0, 1)
    cg, co = g.get_correspondences(), o.get_correspondences()
    for a, b in zip(cg, co):
        assert np.array_equal(a, b)
    assert np.abs(cg[0] - cg[1]).max() > 100                      # lap-to-lap revisits were matched: pose pairs far outside the band
    pg, sg = g.solve(abi.MODE_JOINT, d["init"])
    po, so = o.solve(abi.MODE_JOINT, d["init"])
    assert sg.num_offband_pairs > 0 and sg.linear_iterations > 2 * sg.iterations
    assert sg.pcg_timed_calls == sg.iterations and sg.pcg_timed_iterations == sg.linear_iterations - sg.iterations and sg.pcg_ms > 0
    assert sg.termination == so.termination and sg.iterations == so.iterations
    assert sg.successful_steps == so.successful_steps and sg.unsuccessful_steps == so.unsuccessful_steps
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    dxy, dth = _pdiff(pg, po)
    assert dxy <= 1e-4 and dth <= 1e-4, (dxy, dth)
    print("figure-8 400 x 120 all pairs: %d LM / %d CG iterations, %.1f us per CG iteration, cost %.12g (oracle %.12g), |dxy| %.2e |dth| %.2e"
          % (sg.iterations, sg.linear_iterations - sg.iterations, 1e3 * sg.pcg_ms / max(sg.pcg_timed_iterations, 1), sg.final_cost, so.final_cost, dxy, dth))
    g.close(); o.close()


@pytest.mark.parametrize("band", [1, 2, 3, 4])
@pytest.mark.parametrize("n,m,laps", [(64, 60, 2), (150, 90, 2), (333, 60, 3), (700, 48, 4), (2000, 24, 8)])
def test_persistent_launch_walks_the_path_of_the_launch_per_stage_loop(switches, band, n, m, laps):
    """Same map, same correspondences, same options: the persistent launch (Chronopoulos-Gear recurrence, workgroup roles) and the loop
    of rounds 2-5 (textbook PCG, eight launches per iteration) solve every linear system to the same tolerance: the same LM iterations,
    accepted steps and termination, final costs equal to 1e-9, poses to 1e-6.  Band widths 1-4 (super-blocks of 3, 6, 9 and 12), chains
    from 16 to 2 000 nodes (2 000 poses at a band of 4: 32 top nodes of 12 x 12, the five-slot instantiation of the sweep)."""
    import hitl_slam_amd as H
    d, rows = _fig8(n, m, laps)
    res = []
    corr = None
    for fused in (1, 0):
        switches(pcg_fused=fused)
        opt = H.default_options()
        opt.max_band_poses = band
        opt.timing_level = 2
        g = _load(H.Solver(opt), d, rows)
        if corr is None:
            g.find_correspondences(d["init"], 0)
            corr = g.get_correspondences()
        else:
            g.set_correspondences(*corr)
        p, sm = g.solve(abi.MODE_JOINT, d["init"], allow_failure=True)
        res.append((p, sm))
        g.close()
    (pf, sf), (pl, sl) = res
    assert sf.num_offband_pairs == sl.num_offband_pairs > 0 and sf.band_poses == sl.band_poses <= band
    assert sl.pcg_timed_calls == 0
    # (chains shorter than four super-blocks keep the loop: nothing to spread over workgroups)
    assert sf.pcg_timed_calls == (sf.iterations if (n + sf.band_poses - 1) // sf.band_poses >= 4 else 0)
    assert (sf.termination, sf.iterations, sf.successful_steps, sf.unsuccessful_steps) == (sl.termination, sl.iterations, sl.successful_steps, sl.unsuccessful_steps)
    assert abs(sf.final_cost - sl.final_cost) <= 1e-9 * sl.final_cost
    dxy, dth = _pdiff(pf, pl)
    assert dxy <= 1e-6 and dth <= 1e-6, (dxy, dth)


def test_a_wait_that_runs_out_voids_the_solve_and_the_host_repeats_it(switches):
    """pcg_starve: the launch starts without its top workgroup and with short waits -- what a device too busy to hold every workgroup at once
    looks like.  The chunk workgroups give up, the record in pinned memory says so, the host repeats that LM iteration on the
    launch-per-stage loop and stays there: the result is the loop's result, no error, no hang."""
    import hitl_slam_amd as H
    d, rows = _fig8(150, 90, 2)
    out = []
    corr = None
    for hooks in (dict(pcg_fused=0), dict(pcg_starve=1)):
        switches(**hooks)
        opt = H.default_options()
        opt.timing_level = 2
        g = _load(H.Solver(opt), d, rows)
        if corr is None:
            g.find_correspondences(d["init"], 0)
            corr = g.get_correspondences()
        else:
            g.set_correspondences(*corr)
        out.append(g.solve(abi.MODE_JOINT, d["init"]))
        # a second solve on the same solver: the persistent launch stays off (no second timeout)
        out.append(g.solve(abi.MODE_JOINT, d["init"]))
        g.close()
        H.set_test_switch(None)
    (p0, s0), (p0b, s0b), (p1, s1), (p1b, s1b) = out
    assert s1.pcg_timed_calls == 0 and s1b.pcg_timed_calls == 0
    for p, sm in ((p1, s1), (p1b, s1b)):
        assert (sm.termination, sm.iterations, sm.successful_steps) == (s0.termination, s0.iterations, s0.successful_steps)
        assert sm.final_cost == s0.final_cost and np.array_equal(p, p0)


def test_corridor_all_pairs_dense_off_band_rows():
    """One straight corridor searched all pairs: ~90 off-band blocks per pose (every pose sees most of the others), the densest rows the
    product kernel meets; against the oracle."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    prob = Problem(240, 96, ppf=240, groups=2)
    g, o = prob.load(H.Solver()), prob.load(O.OracleSolver())
    g.find_correspondences(prob.init, 0)
    o.find_correspondences(prob.init, 0, 1)
    for a, b in zip(g.get_correspondences(), o.get_correspondences()):
        assert np.array_equal(a, b)
    pg, sg = g.solve(abi.MODE_JOINT, prob.init)
    po, so = o.solve(abi.MODE_JOINT, prob.init)
    print("corridor 240 x 96 all pairs: %d off-band block pairs (%.1f per pose), %d LM / %d CG iterations" % (
        sg.num_offband_pairs, sg.num_offband_pairs / prob.n, sg.iterations, sg.linear_iterations - sg.iterations))
    assert sg.num_offband_pairs > 2 * prob.n
    assert sg.termination == so.termination and sg.iterations == so.iterations
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    dxy, dth = _pdiff(pg, po)
    assert dxy <= 1e-4 and dth <= 1e-4
    g.close(); o.close()


def test_configs2_at_full_size_both_forms_walk_one_path(switches):
    """BASELINE configs[2] at bench.py --workload c3's size (1 000 poses x 360 points, 4 laps, 5 + 3 constraint groups, all pairs: the
    oracle's dense factorisation would take 40 s here; it is held to the oracle at 400 x 120 above): the persistent launch and the
    launch-per-stage loop take the same LM iterations to the same cost (1e-9) and poses (1e-6), every linear solve of the
    former is ONE launch, and the map's hub poses (more than 200 off-band partners) are there."""
    import hitl_slam_amd as H
    d, rows = _fig8(1000, 360, 4)
    res, corr = [], None
    for fused in (1, 0):
        switches(pcg_fused=fused)
        opt = H.default_options()
        opt.timing_level = 2
        g = _load(H.Solver(opt), d, rows)
        if corr is None:
            g.find_correspondences(d["init"], 0)
            corr = g.get_correspondences()
        else:
            g.set_correspondences(*corr)
        res.append(g.solve(abi.MODE_JOINT, d["init"]))
        g.close()
    (pf, sf), (pl, sl) = res
    partners = np.bincount(np.concatenate([corr[0][np.abs(corr[0] - corr[1]) > sf.band_poses], corr[1][np.abs(corr[0] - corr[1]) > sf.band_poses]]), minlength=1000)
    assert partners.max() > 200 and sf.num_offband_pairs > 10000
    assert sf.pcg_timed_calls == sf.iterations and sl.pcg_timed_calls == 0
    assert (sf.termination, sf.iterations, sf.successful_steps, sf.unsuccessful_steps) == (sl.termination, sl.iterations, sl.successful_steps, sl.unsuccessful_steps)
    assert abs(sf.final_cost - sl.final_cost) <= 1e-9 * sl.final_cost and sf.final_cost < sf.initial_cost
    dxy, dth = _pdiff(pf, pl)
    assert dxy <= 1e-6 and dth <= 1e-6, (dxy, dth)
    print("configs[2] size: %d LM / %d CG iterations (loop: %d), %.1f us per CG iteration, %d off-band block pairs, hub pose: %d directed partners" % (
        sf.iterations, sf.linear_iterations - sf.iterations, sl.linear_iterations - sl.iterations, 1e3 * sf.pcg_ms / max(sf.pcg_timed_iterations, 1),
        sf.num_offband_pairs, int(partners.max())))


def test_configs2_at_full_size_against_the_oracles_exact_solve():
    """BASELINE configs[2] at bench.py --workload c3's size against the oracle (its search walks 15 M trees, its solver factorises a dense
    3 000 x 3 000 band per LM iteration: about a minute of CPU): identical correspondence sets (17 k directed pairs, 2.09 M matches), the
    same LM path (iterations, accepted and rejected steps, termination), cost to 1e-8, poses within 1e-4 m / 1e-4 rad."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d, rows = _fig8(1000, 360, 4)
    g, o = _load(H.Solver(), d, rows), _load(O.OracleSolver(), d, rows)
    g.find_correspondences(d["init"], 0)
    o.find_correspondences(d["init"], 0, 1)
    cg, co = g.get_correspondences(), o.get_correspondences()
    for a, b in zip(cg, co):
        assert np.array_equal(a, b)
    assert len(cg[0]) > 15000 and len(cg[3]) > 2000000
    pg, sg = g.solve(abi.MODE_JOINT, d["init"])
    po, so = o.solve(abi.MODE_JOINT, d["init"])
    assert sg.termination == so.termination and sg.iterations == so.iterations
    assert sg.successful_steps == so.successful_steps and sg.unsuccessful_steps == so.unsuccessful_steps
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    dxy, dth = _pdiff(pg, po)
    assert dxy <= 1e-4 and dth <= 1e-4, (dxy, dth)
    print("configs[2] size against the oracle: %d pairs, %d matches, %d LM iterations on both sides, cost %.12g (oracle %.12g), |dxy| %.2e |dth| %.2e"
          % (len(cg[0]), len(cg[3]), sg.iterations, sg.final_cost, so.final_cost, dxy, dth))
    g.close(); o.close()


def test_ragged_map_with_empty_scans_on_the_pcg_path(switches):
    """Scans of different lengths, two poses without any point (rows of the system that only odometry and the band touch: identity parts
    of the product items), a few NaN returns: the all-pairs sets equal the oracle's, the persistent launch, the launch-per-stage loop
    and the oracle's exact solve agree."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d, rows = _fig8(220, 100, 2)
    rng = np.random.default_rng(3)
    off = np.asarray(d["off"]).copy()
    cnt = np.diff(off).copy()
    keep = np.ones(int(off[-1]), bool)
    for i in range(220):                                   # ragged: drop 0-40 % of every scan's points; two scans entirely
        drop = rng.random(cnt[i]) < (1.0 if i in (10, 57) else rng.uniform(0.0, 0.4))
        keep[off[i]:off[i + 1]] = ~drop
        cnt[i] = int((~drop).sum())
    d2 = dict(d)
    d2["off"] = np.concatenate([[0], np.cumsum(cnt)]).astype(off.dtype)
    for k in ("px", "py", "nx", "ny"):
        d2[k] = np.asarray(d[k], np.float32)[keep].copy()
    d2["px"][rng.integers(0, len(d2["px"]), 25)] = np.float32(np.nan)
    assert cnt[10] == 0 and cnt[57] == 0 and cnt.min() == 0 and cnt.max() > 80
    res, corr = [], None
    for fused in (1, 0):
        switches(pcg_fused=fused)
        opt = H.default_options()
        opt.timing_level = 2
        g = _load(H.Solver(opt), d2, rows)
        g.find_correspondences(d2["init"], 0)
        corr = g.get_correspondences()
        res.append(g.solve(abi.MODE_JOINT, d2["init"]))
        g.close()
    o = _load(O.OracleSolver(), d2, rows)
    o.find_correspondences(d2["init"], 0, 1)
    for a, b in zip(corr, o.get_correspondences()):
        assert np.array_equal(a, b)
    po, so = o.solve(abi.MODE_JOINT, d2["init"])
    o.close()
    (pf, sf), (pl, sl) = res
    assert sf.num_offband_pairs > 0 and sf.pcg_timed_calls == sf.iterations and sl.pcg_timed_calls == 0
    for sm, p in ((sf, pf), (sl, pl)):
        assert (sm.termination, sm.iterations, sm.successful_steps) == (so.termination, so.iterations, so.successful_steps)
        assert abs(sm.final_cost - so.final_cost) <= 1e-8 * so.final_cost
        dxy, dth = _pdiff(p, po)
        assert dxy <= 1e-4 and dth <= 1e-4, (dxy, dth)


@pytest.mark.parametrize("mode", [abi.MODE_STF_ONLY, abi.MODE_JOINT])
@pytest.mark.parametrize("kinds", [(2, 4), (5, 6), (4, 6)])
def test_modes_and_constraint_kinds_on_the_pcg_path(mode, kinds):
    """Mode B (scan-to-scan factors only: JointOptimization.cpp:1140-1224) and mode C, human rows of the kinds the reference builds
    (line-segment, colinear, perpendicular, parallel: residual_functors.h:1305-1411), on an all-pairs map whose linear solves run as
    the persistent launch: the oracle's LM path, poses within 1e-4 m / 1e-4 rad, cost to 1e-8 in mode C.  Mode B has no odometry
    and no human rows: its first systems are ill-conditioned (~190 CG iterations against ~20), the oracle's LM runs to its
    100-iteration cap, and ANY iterative solve held to a relative residual of 1e-12 lands a condition number away from the exact
    Cholesky's step (the launch-per-stage loop too: tools/dev/pcg_true_residual.py shows recurrence and true residual agreeing to
    three digits) -- there the comparison is made at hitl_options.linear_solver_tolerance = 1e-14 and a cost tolerance of 1e-5."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d = figure8.make(180, 90, seed=13, laps=2)
    raw = figure8.constraints(d["gt"], 4, 4)
    raw = [((kinds[0] if r[0] == 2 else kinds[1]),) + tuple(r[1:]) for r in raw]          # the same anchors and targets, other row kinds
    rows = hc_rows(raw)
    opt = H.default_options()
    opt.timing_level = 2
    if mode == abi.MODE_STF_ONLY:
        opt.linear_solver_tolerance = 1e-14
    g, o = _load(H.Solver(opt), d, rows), _load(O.OracleSolver(), d, rows)
    g.find_correspondences(d["init"], 0)
    o.find_correspondences(d["init"], 0, 1)
    for a, b in zip(g.get_correspondences(), o.get_correspondences()):
        assert np.array_equal(a, b)
    pg, sg = g.solve(mode, d["init"], allow_failure=True)
    po, so = o.solve(mode, d["init"], allow_failure=True)
    assert sg.num_offband_pairs > 0 and sg.pcg_timed_calls == sg.iterations
    assert (sg.termination, sg.iterations, sg.successful_steps, sg.unsuccessful_steps) == (so.termination, so.iterations, so.successful_steps, so.unsuccessful_steps)
    assert abs(sg.final_cost - so.final_cost) <= (1e-5 if mode == abi.MODE_STF_ONLY else 1e-8) * max(so.final_cost, 1e-300)
    dxy, dth = _pdiff(pg, po)
    assert dxy <= 1e-4 and dth <= 1e-4, (dxy, dth)
    g.close(); o.close()
