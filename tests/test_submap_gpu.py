"""Pose-range submaps (the N > 1 path) on the real HIP solver: two ranks, one submap each.
ppf=30: the cut falls between two floors (only the odometry edge straddles it).  ppf=120: ONE
corridor cut in the middle, so scan-to-scan pairs within the matching window straddle the cut
and the 6-pose halos matter.  The exchange runs through the host-buffer all-gather callback
over gloo, so both ranks can share the single GPU of the test box.  The distributed result must
equal the oracle's GLOBAL solve."""
import json
import os
import subprocess
import sys

import numpy as np
import pytest

from conftest import Problem


def _free_port():
    """A port nobody is listening on (rendezvous of the worker processes on 127.0.0.1)."""
    import socket
    with socket.socket(socket.AF_INET, socket.SOCK_STREAM) as sk:
        sk.bind(("127.0.0.1", 0))
        return sk.getsockname()[1]

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


# `form`: how the ranks' linear solves must have run (hitl_summary.submap_linear_solver): 2 = separators (every rank but the last
# owns a multiple of the pose band: 60-pose ranks with a band of 3 or 1, the cut lists 0,39,81,120 / 0,42,72,120), 1 = interface
# columns (40-pose ranks with a band of 3, uneven cuts, or HITL_DIST_COLUMNS=1 on a partition the separators would take)
@pytest.mark.parametrize("mode,ppf,world,moments,cuts,windows,form",
                         [(0, 30, 2, 0, None, None, 2), (2, 30, 2, 0, None, None, 2), (2, 120, 2, 0, None, None, 2), (0, 120, 2, 0, None, None, 2),
                          (2, 120, 3, 0, None, None, 1), (2, 40, 3, 0, None, None, 1), (2, 120, 2, 1, None, None, 2),
                          (2, 120, 3, 0, "0,23,83,120", None, 1),      # uneven submaps: 23, 60 and 37 poses
                          # rank 0 = the first floor, matched within +-1 pose only: its own band is one pose wide while the cut
                          # between ranks 1 and 2 (inside the second floor) is crossed by pairs three poses long.  Every rank must
                          # widen its band to the interface of the whole partition (round 2: rank 0 alone refused and the others hung)
                          (2, 40, 3, 0, "0,40,70,120", "1,3,3", 1),
                          # round 4, the separator form on three ranks: uneven block-aligned submaps (the middle rank has a separator
                          # pair on both sides), mode A on them (band of one pose), the rank with the narrow band widening to the
                          # common one, and the column form forced on a partition the separators would take
                          (2, 120, 3, 0, "0,39,81,120", None, 2), (0, 120, 3, 0, "0,39,81,120", None, 2), (2, 42, 3, 0, "0,42,72,120", "1,3,3", 2),
                          (2, 120, 2, 0, None, None, -1),
                          # HITL_DIST_COLUMNS is a per-process switch: set on rank 1 only, BOTH ranks must fall back to the columns
                          # (the verdict travels in prepare's exchange; a rank deciding alone would leave the other in a collective)
                          (2, 120, 2, 0, None, None, -2)])
def test_submaps_equal_the_global_solve(tmp_path, mode, ppf, world, moments, cuts, windows, form):
    from oracle import hitl_oracle as O
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HITL_TEST_STF_MOMENTS=str(moments))
    if cuts:
        env["HITL_TEST_CUTS"] = cuts
    if windows:
        env["HITL_TEST_WINDOWS"] = windows
    if form == -1:
        env["HITL_DIST_COLUMNS"] = "1"
        form = 1
    if form == -2:
        env["HITL_TEST_COLUMNS_RANK"] = "1"
        form = 1
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, str(mode), str(ppf)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(world)]

    prob = Problem(120, 48, ppf=ppf, groups=2, kind=0)
    o = prob.load(O.OracleSolver())
    o.find_correspondences(prob.init, 3, 1)
    ops = o.get_correspondences()
    if windows:      # the global problem of the per-rank windows: inside rank r's range only pairs within its window (6 slots <= cap 6: the cap never binds)
        cl, wl = [int(c) for c in cuts.split(",")], [int(w) for w in windows.split(",")]
        rank_of = np.searchsorted(np.array(cl[1:]), np.arange(prob.n), side="right")
        keep = np.array([not (rank_of[a] == rank_of[b] and abs(int(a) - int(b)) > wl[rank_of[a]]) for a, b in zip(ops[0], ops[1])])
        assert 0 < keep.sum() < len(keep)
        lens = (ops[2][1:] - ops[2][:-1])[keep]
        sel = np.concatenate([np.arange(ops[2][p], ops[2][p + 1]) for p in np.nonzero(keep)[0]])
        ops = (ops[0][keep], ops[1][keep], np.concatenate([[0], np.cumsum(lens)]).astype(np.int64), ops[3][sel], ops[4][sel])
        o.set_correspondences(*ops)
    # the pairs reported by the ranks owning their source pose tile the global set exactly
    src = np.concatenate([np.array(x["pairs"][0], np.int32) for x in res])
    tgt = np.concatenate([np.array(x["pairs"][1], np.int32) for x in res])
    assert np.array_equal(src, ops[0]) and np.array_equal(tgt, ops[1])
    if ppf == 120:
        cut = res[1]["begin"]
        assert np.any((ops[0] < cut) != (ops[1] < cut))        # pairs really straddle the cut
    if world == 3 and not cuts:
        assert res[1]["begin"] == 40 and res[1]["end"] == 80   # the middle rank has halos on both sides
    if cuts:
        assert [x["begin"] for x in res] + [res[-1]["end"]] == [int(c) for c in cuts.split(",")]
    assert sum(x["n_matches"] for x in res) == len(ops[3])
    # observation scans on the submap ranks == the same calls of the oracle on the whole map, restricted to the rank's
    # resident poses [lo, hi) (bit for bit: fp32, host trigonometry)
    M = 48
    owx, owy = o.transform_scans(prob.init)
    for x in res:
        sc = x["scans"]
        sl = slice(sc["lo"] * M, sc["hi"] * M)
        assert np.array_equal(np.array(sc["wx"], np.float32), owx[sl]) and np.array_equal(np.array(sc["wy"], np.float32), owy[sl])
        seen = o.verify_points(prob.init, np.array(sc["probe"], np.float32), 0.05)
        assert sc["seen"][0] == 1 and sc["seen"][1] == 1 and sc["seen"][2] == 0 and seen[0] == 1 and seen[2] == 0
        oc, om = o.observation_sets(prob.init, np.array(sc["seg"], np.float32), 0.03)
        assert np.array_equal(np.array(sc["counts"], np.int32), oc[sc["lo"]:sc["hi"]]) and sc["mask_sum"] == int(om[sl].sum())
    # endpoint fit (EMInput::AutomaticEndpointAdjustment) over the ranks == the oracle's fit on the whole map: same inlier
    # set (gathered from the owners in pose order), same passes and LM path; identical on every rank
    fo = o.fit_segments(prob.init, np.array(res[0]["fit"]["clicks"], np.float32).reshape(1, 4))
    for x in res:
        f = x["fit"]
        assert f["clicks"] == res[0]["fit"]["clicks"]
        assert f["inliers"] == fo[1].tolist() and f["passes"] == fo[2].tolist() and f["its"] == fo[3].tolist()
        assert np.abs(np.array(f["seg"], np.float32) - fo[0]).max() <= 2e-6
        assert f["seg"] == res[0]["fit"]["seg"]
    assert fo[1][0] > 0
    if ppf == 120:      # one corridor: the wall piece is seen from both sides of the first cut, so the inliers come from two ranks
        cnt, _ = o.observation_sets(prob.init, np.r_[fo[0][0], fo[0][0]].astype(np.float32), 0.03)
        seen_by = np.nonzero(cnt.reshape(-1, 2)[:, 0] > 0)[0]
        cut = res[1]["begin"]
        assert seen_by.min() < cut <= seen_by.max()
    # evaluate on the ranks (cost + gradient at the initial poses): the global cost on every rank, each rank's owned rows
    oc, og, _, _ = o.linearize(mode, prob.init, prob.init.astype(np.float64))
    for x in res:
        ev = x["evaluate"]
        assert abs(ev["cost"] - oc) <= 1e-11 * oc
        gr = np.array(ev["grad"])
        lo3, hi3 = 3 * x["begin"], 3 * x["end"]
        assert np.abs(gr[lo3:hi3] - og[lo3:hi3]).max() <= 1e-9 * np.abs(og).max()
        assert not gr[:lo3].any() and not gr[hi3:].any()
    # pose covariances on the ranks: the 3x3 blocks of the GLOBAL inverse over each rank's owned poses (the rank's own
    # selected inversion minus the rank-2m correction through its interface columns) == the oracle's dense inverse
    So = o.pose_covariances(mode, prob.init)
    for x in res:
        Sg = np.array(x["cov"])
        b, e = x["begin"], x["end"]
        scale = np.abs(So[b:e]).max(axis=(1, 2))[:, None, None] + 1e-300
        assert (np.abs(Sg[b:e] - So[b:e]) / scale).max() < 1e-6, (np.abs(Sg[b:e] - So[b:e]) / scale).max()
        assert not Sg[:b].any() and not Sg[e:].any()
    po, so = o.solve(mode, prob.init)
    for x in res:
        sm = x["summary"]
        assert sm["submap_linear_solver"] == form, sm["submap_linear_solver"]
        assert sm["termination"] == so.termination and sm["iterations"] == so.iterations
        assert sm["successful_steps"] == so.successful_steps
        assert abs(sm["final_cost"] - so.final_cost) <= 1e-8 * so.final_cost
        assert abs(sm["initial_cost"] - so.initial_cost) <= 1e-10 * so.initial_cost
        assert sm["num_residuals"] == so.num_residuals and sm["num_hc_residuals"] == so.num_hc_residuals
        got = np.array(x["poses"], np.float32)[x["begin"]:x["end"]]
        assert np.abs(got[:, :2] - po[x["begin"]:x["end"], :2]).max() <= 1e-4
        assert np.abs(got[:, 2] - po[x["begin"]:x["end"], 2]).max() <= 1e-4


@pytest.mark.parametrize("world", [2, 3])
def test_mode_b_on_submap_ranks_reaches_the_global_cost_level(tmp_path, world):
    """Mode B (PostHumanOptimization as written: scan-to-scan factors only, pose 0 held) on submap ranks, one corridor cut inside
    (pairs straddle the cuts).  The STF-only problem is ill-conditioned and does not converge in 100 iterations; its
    accept / reject path depends on the rounding of the linear solver even between two dense solvers of the same normal
    equations (tests/test_golden.py), so ranks and the single-GPU oracle are held to what CAN agree: the
    cost and gradient at the start exactly, the final cost level (2 %; measured 5e-4), poses to 3e-2 -- not the iteration count (75 on one GPU,
    100 / 64 on 2 / 3 ranks when this was first tried in round 2) and hence not the termination flag."""
    from oracle import hitl_oracle as O
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, "1", "120"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(world)]
    prob = Problem(120, 48, ppf=120, groups=2, kind=0)
    o = prob.load(O.OracleSolver())
    o.find_correspondences(prob.init, 3, 1)
    oc, og, _, _ = o.linearize(1, prob.init, prob.init.astype(np.float64))
    po, so = o.solve(1, prob.init)
    for x in res:
        assert abs(x["evaluate"]["cost"] - oc) <= 1e-11 * oc
        gr = np.array(x["evaluate"]["grad"])
        assert np.abs(gr[3 * x["begin"]:3 * x["end"]] - og[3 * x["begin"]:3 * x["end"]]).max() <= 1e-9 * np.abs(og).max()
        sm = x["summary"]
        assert abs(sm["initial_cost"] - so.initial_cost) <= 1e-10 * so.initial_cost
        got = np.array(x["poses"], np.float32)[x["begin"]:x["end"]]
        print("\nmode B on %d ranks, rank %d: term %d iters %d cost %.6g | one GPU (oracle): term %d iters %d cost %.6g | max pose difference %.3g"
              % (world, x["rank"], sm["termination"], sm["iterations"], sm["final_cost"], so.termination, so.iterations, so.final_cost,
                 np.abs(got - po[x["begin"]:x["end"]]).max()))
        assert sm["termination"] in (0, 1) and sm["final_cost"] < sm["initial_cost"]
        assert abs(sm["final_cost"] - so.final_cost) <= 2e-2 * so.final_cost
        assert np.abs(got - po[x["begin"]:x["end"]]).max() <= 3e-2      # along the weakly observed corridor direction (1.5e-2 on 2 ranks, 9e-4 on 3)


@pytest.mark.parametrize("world,window,nper,halo", [(2, 1, 60, 2), (2, 2, 60, 4), (3, 2, 40, 4), (2, 4, 60, 8), (3, 4, 40, 8),
                                                    (3, 3, 30, -1), (2, 3, 60, 3)])
def test_separator_form_at_other_band_widths(tmp_path, world, window, nper, halo):
    """Round 4: the separator form with super-blocks of 1, 2 and 4 poses (B = 3, 6, 12; the other cases run B = 9 and, in mode A,
    B = 3): ranks whose pose counts are multiples of the band inside one corridor, so the solve AND the pose covariances on the
    ranks take it; both must equal the oracle's global solve / dense global inverse.  Also: three ranks holding REPLICATED scans
    (halo < 0: every rank has the whole map resident), and halos exactly as deep as the band."""
    from oracle import hitl_oracle as O
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HITL_TEST_WINDOW=str(window), HITL_TEST_HALO=str(halo),
               HITL_TEST_NPER=str(nper))
    n = nper * world
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, "2", str(n)]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=600)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(world)]
    prob = Problem(n, 48, ppf=n, groups=2, kind=0)
    o = prob.load(O.OracleSolver())
    o.find_correspondences(prob.init, window, 1)
    ops = o.get_correspondences()
    assert np.abs(ops[0] - ops[1]).max() == window
    So = o.pose_covariances(2, prob.init)
    po, so = o.solve(2, prob.init)
    for x in res:
        sm = x["summary"]
        assert sm["submap_linear_solver"] == 2 and sm["band_poses"] == window
        assert sm["termination"] == so.termination and sm["iterations"] == so.iterations and sm["successful_steps"] == so.successful_steps
        assert abs(sm["final_cost"] - so.final_cost) <= 1e-8 * so.final_cost
        b, e = x["begin"], x["end"]
        got = np.array(x["poses"], np.float32)[b:e]
        assert np.abs(got - po[b:e]).max() <= 1e-4
        Sg = np.array(x["cov"])
        scale = np.abs(So[b:e]).max(axis=(1, 2))[:, None, None] + 1e-300
        assert (np.abs(Sg[b:e] - So[b:e]) / scale).max() < 1e-6


@pytest.mark.parametrize("inject, mode, cuts", [("fail_prepare=1", 2, None), ("fail_prepare=1", 0, "0,39,81,120"), ("nan_pose", 2, None),
                                                ("poison_step_from=2", 2, None), ("poison_step_from=1", 0, "0,39,81,120")])
def test_a_failure_on_one_rank_fails_every_rank(tmp_path, inject, mode, cuts):
    """one rank of a submap solve fails -- (a) in the rank-local part of prepare (an allocation failure, injected by
    the library's test hook on rank 1 only), (b) inside the LM loop (rank 1 hands in a NaN at one of its own poses; rank 1's steps
    stop being finite from some iteration on) -- and EVERY rank returns a non-zero status within the timeout: the verdicts travel
    in prepare's exchange and in the scalars that end every LM iteration, so nobody is left waiting in a collective.  Two and three
    ranks share the test box's GPU over gloo; a hung rank would run into the subprocess timeout."""
    world = 3 if cuts else 2
    out = str(tmp_path / "res")
    env = dict(os.environ, MASTER_ADDR="127.0.0.1", HSA_ENABLE_IPC_MODE_LEGACY="0", HITL_TEST_EXPECT_FAIL="1", HITL_TEST_INJECT=inject,
               HITL_TEST_INJECT_RANK="1")
    if cuts:
        env["HITL_TEST_CUTS"] = cuts
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(world), "--master-addr", "127.0.0.1",
           "--master-port", str(_free_port()), os.path.join(ROOT, "tests", "_submap_worker.py"), out, str(mode), "120"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=300)
    assert r.returncode == 0, r.stdout[-2000:] + r.stderr[-4000:]
    res = [json.load(open("%s.%d" % (out, k))) for k in range(world)]
    assert all(x["status"] != 0 for x in res), res
    if inject.startswith("fail_prepare"):
        assert "injected" in res[1]["error"] and all("another rank failed" in res[k]["error"] for k in range(world) if k != 1), res
    else:                                    # the LM loop ends with FAILURE on every rank alike (status 7 = HITL_ERR_SOLVER_FAILURE)
        assert len({x["status"] for x in res}) == 1 and "FAILURE" in res[0]["error"], res
