"""GPU parity of the correction cycle around the solve through the C ABI, against the CPU
restatement: fit_segments, order_user_input, apply_explicit_correction, backprop, normalize_angles, and one whole
HitLSLAM::Run.  Index and fp32 pose arithmetic is compared bit for bit; the only tolerance is 4 ulp on the one field
whose value goes through the device's atan2f/sinf/cosf (delta_angle)."""
import time

import numpy as np
import pytest

from hitl_slam_amd import abi
from tests.cycle_scenario import Scenario
from tests.test_cycle_oracle import _chain

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def sc():
    return Scenario()


@pytest.fixture(scope="module")
def both(sc):
    from hitl_slam_amd import Solver
    from oracle import hitl_oracle as O
    g, o = Solver(), O.OracleSolver()
    sc.prob.load(g)
    sc.prob.load(o)
    yield g, o
    g.close()
    o.close()


def test_fit_segments_parity(both, sc):
    g, o = both
    for pts in (sc.selected.reshape(2, 4), Scenario(first_piece=(10.5, 17.5)).selected.reshape(2, 4)):
        sg, ig, pg, tg = g.fit_segments(sc.poses, pts)
        so, io, po, to = o.fit_segments(sc.poses, pts)
        assert np.array_equal(ig, io) and np.array_equal(pg, po) and np.array_equal(tg, to)
        # same inliers, same LM path; endpoints are fp64 values rounded to fp32: allow the last bit
        assert np.abs(sg - so).max() <= 2e-6, (sg, so)
    # no inliers at all
    sg, ig, pg, tg = g.fit_segments(sc.poses, [[500.0, 500.0, 503.0, 496.0]])
    so, io, po, to = o.fit_segments(sc.poses, [[500.0, 500.0, 503.0, 496.0]])
    assert ig[0] == 0 and np.array_equal(sg, so) and np.array_equal(pg, po)
    with pytest.raises(Exception):
        g.fit_segments(sc.poses, [[1.0, 2.0, 1.0, 2.0]])


@pytest.mark.parametrize("alpha", [-0.5, 0.7, 2.0, -2.6])
def test_fit_segments_parity_on_a_rotated_world(both, sc, alpha):
    """Walls that are not axis-aligned: SegFitEM starts from acos(|dx| / length), which drops the sign of the slope, so
    for a falling wall the fit has to swing the segment through a large angle (or settles in whatever minimum the
    reference would).  The map is rotated by moving the poses; the robot-frame scans stay as they are."""
    g, o = both
    c, s_ = np.cos(alpha), np.sin(alpha)
    R = np.array([[c, -s_], [s_, c]])
    poses = sc.poses.astype(np.float64).copy()
    poses[:, :2] = poses[:, :2] @ R.T
    poses[:, 2] += alpha
    poses = poses.astype(np.float32)
    wx, wy = o.transform_scans(poses)
    W = np.c_[wx, wy]
    clicks = (sc.selected.reshape(4, 2).astype(np.float64) @ R.T)
    # snap every click to the nearest point of the rotated map, like a user would click
    clicks = np.array([W[np.argmin(((W - q) ** 2).sum(1))] for q in clicks], np.float32)
    sg, ig, pg, tg = g.fit_segments(poses, clicks.reshape(2, 4))
    so, io, po, to = o.fit_segments(poses, clicks.reshape(2, 4))
    assert np.array_equal(ig, io) and np.array_equal(pg, po) and np.array_equal(tg, to), (ig, io, pg, po, tg, to)
    assert np.abs(sg - so).max() <= 4e-6, (sg, so)


@pytest.mark.parametrize("seed", [20260111, 20260112, 20260113])
def test_full_cycle_parity_other_maps(seed):
    from hitl_slam_amd import Solver
    from oracle import hitl_oracle as O
    sc2 = Scenario(n_poses=100, obs=96, seed=seed)
    g, o = Solver(), O.OracleSolver()
    sc2.prob.load(g)
    sc2.prob.load(o)
    rg = g.run_correction(4, sc2.selected, sc2.poses, sc2.cov, [])            # the default: re-projection included, as in the reference's Run
    ro = o.run_correction(4, sc2.selected, sc2.poses, sc2.cov, [], reproject=False)
    assert rg["status"] == ro["status"] == "ok" and "world" not in ro and "world" in rg
    assert rg["order"]["bounds"] == ro["order"]["bounds"] and np.array_equal(rg["order"]["corrected"], ro["order"]["corrected"])
    d = rg["poses"].astype(np.float64) - ro["poses"]
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    assert np.abs(d[:, :2]).max() < 1e-4 and np.abs(d[:, 2]).max() < 1e-4
    # the display clouds of the cycle's last step (transformPointCloudsToWorldFrame): page-locked buffers the Solver keeps,
    # bit for bit the pageable-memory form at the same poses
    wx, wy = g.transform_scans(rg["poses"])
    assert np.array_equal(rg["world"][0].view(np.uint32), wx.view(np.uint32)) and np.array_equal(rg["world"][1].view(np.uint32), wy.view(np.uint32))
    # a LARGER map loaded into the same solver: the page-locked buffers are sized by the map they were allocated for (ADVICE r4:
    # they used to be reused unconditionally -- a heap overflow); set_map drops them, and what world_clouds returned earlier are
    # copies the caller owns (the views of copy=False -- what run_correction hands out -- die with the buffers: copied here first)
    old = np.array(rg["world"][0])
    from tests.conftest import Problem
    big = Problem(130, 128, seed=seed + 7, groups=0)
    big.load(g)
    assert g._world_buf is None and np.array_equal(old.view(np.uint32), wx.view(np.uint32))
    wx3, wy3 = g.world_clouds(big.init)
    ex3, ey3 = g.transform_scans(big.init)
    assert wx3.size == ex3.size > wx.size and np.array_equal(wx3.view(np.uint32), ex3.view(np.uint32)) and np.array_equal(wy3.view(np.uint32), ey3.view(np.uint32))
    g.close(); o.close()
    assert np.array_equal(wx3.view(np.uint32), ex3.view(np.uint32))            # still the caller's after close() freed the staging buffers


@pytest.mark.parametrize("ctype", [2, 4, 5, 6])
def test_explicit_correction_parity(both, sc, ctype):
    g, o = both
    corrected = np.r_[np.arange(66, 85), np.arange(90, 95)].astype(np.int32)     # two groups: only the first moves explicitly
    anchor = np.arange(0, 34, dtype=np.int32)
    pg, cg, rg = g.apply_explicit_correction(ctype, sc.selected, sc.poses, corrected, anchor)
    po, co, ro = o.apply_explicit_correction(ctype, sc.selected, sc.poses, corrected, anchor)
    assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))
    assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))
    assert rg.size == ro.size == corrected.size * anchor.size
    for f in ("type", "constrained_pose_id", "anchor_pose_id"):
        assert np.array_equal(rg[f], ro[f])
    for f in ("delta_parallel", "delta_perpendicular", "relative_penalty_dir"):
        assert np.array_equal(rg[f].view(np.uint32), ro[f].view(np.uint32)), f
    ulp = np.spacing(np.abs(ro["delta_angle"]).astype(np.float32)).astype(np.float64)
    assert (np.abs(rg["delta_angle"].astype(np.float64) - ro["delta_angle"]) <= 4 * ulp + 1e-12).all()   # three libm calls deep
    for t in (0, 1, 3):
        with pytest.raises(Exception):
            g.apply_explicit_correction(t, sc.selected, sc.poses, corrected, anchor)


@pytest.mark.parametrize("n,lo,hi", [(400, 37, 352), (300, 0, 255), (300, 3, 259), (300, 2, 259), (700, 10, 11), (700, 5, 699),
                                     (3000, 100, 2900)])
def test_backprop_bit_exact(both, n, lo, hi):
    g, o = both
    poses, cov = _chain(n, seed=n + lo)
    corr = np.array([0.8, -0.5, 0.12], np.float32)
    pg, cg = g.backprop(poses, cov, lo, hi, corr)
    po, co = o.backprop(poses, cov, lo, hi, corr)
    assert np.array_equal(pg.view(np.uint32), po.view(np.uint32)), np.abs(pg - po).max()
    assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))
    assert not np.array_equal(pg, poses)


def test_backprop_no_room_and_bad_bounds(both):
    g, _ = both
    poses, cov = _chain(50)
    p, c = g.backprop(poses, cov, 9, 9, [1, 1, 0.1])
    assert np.array_equal(p, poses) and np.array_equal(c, cov)
    with pytest.raises(Exception):
        g.backprop(poses, cov, 3, 50, [1, 1, 0.1])


def test_backprop_long_span_bit_exact_and_time(both):
    """20 000 poses between the bounds: 2e8 sequential fp32 updates in the reference's loops."""
    g, o = both
    n = 20100
    poses, cov = _chain(n, seed=11)
    corr = np.array([2.5, -1.5, 0.3], np.float32)
    g.backprop(poses, cov, 50, 20050, corr)
    t0 = time.perf_counter()
    pg, cg = g.backprop(poses, cov, 50, 20050, corr)
    t1 = time.perf_counter()
    po, co = o.backprop(poses, cov, 50, 20050, corr)
    t2 = time.perf_counter()
    print("backprop span 20000: gpu %.1f ms (incl. transfers), cpu restatement %.1f ms" % (1e3 * (t1 - t0), 1e3 * (t2 - t1)))
    assert np.array_equal(pg.view(np.uint32), po.view(np.uint32))
    assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))


def test_normalize_angles_parity(both, sc):
    g, o = both
    p = sc.poses.copy()
    p[:, 2] += np.linspace(-20, 20, sc.n).astype(np.float32)
    assert np.array_equal(g.normalize_angles(p).view(np.uint32), o.normalize_angles(p).view(np.uint32))


@pytest.mark.parametrize("ctype", [2, 4, 6])
def test_full_cycle_parity(both, sc, ctype):
    g, o = both
    rg = g.run_correction(ctype, sc.selected, sc.poses, sc.cov, [])
    ro = o.run_correction(ctype, sc.selected, sc.poses, sc.cov, [])
    assert rg["status"] == ro["status"] == "ok"
    assert rg["order"]["bounds"] == ro["order"]["bounds"]
    assert np.array_equal(rg["order"]["corrected"], ro["order"]["corrected"]) and np.array_equal(rg["order"]["anchor"], ro["order"]["anchor"])
    assert np.array_equal(rg["cov9"].view(np.uint32), ro["cov9"].view(np.uint32))
    assert rg["summary"].termination == ro["summary"].termination
    d = rg["poses"].astype(np.float64) - ro["poses"]
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    assert np.abs(d[:, :2]).max() < 1e-4 and np.abs(d[:, 2]).max() < 1e-4      # the north-star pose tolerance
    # a second correction in the same session stacks on the first (human_constraints_ grows; undo = drop the last group)
    sc2 = Scenario(first_piece=(10.5, 17.5))
    rg2 = g.run_correction(4, sc2.selected, rg["poses"], rg["cov9"], rg["constraint_groups"])
    ro2 = o.run_correction(4, sc2.selected, ro["poses"], ro["cov9"], ro["constraint_groups"])
    assert rg2["status"] == ro2["status"]
    if rg2["status"] == "ok":
        assert len(rg2["constraint_groups"]) == 2
        d = rg2["poses"].astype(np.float64) - ro2["poses"]
        d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
        assert np.abs(d[:, :2]).max() < 2e-4 and np.abs(d[:, 2]).max() < 2e-4
    # input that is not on the map is refused
    bad = sc.selected.copy()
    bad[1] += 40.0
    assert g.run_correction(ctype, bad, sc.poses, sc.cov, [])["status"] == "not verified"


def test_full_cycle_with_the_joint_observation_solve(both, sc):
    """The same cycle ending in mode C (odometry + human rows + scan-to-scan factors matched at the corrected poses):
    what HitLSLAM::Run would do with PostHumanOptimization enabled (JointOptimization.cpp:1353-1373)."""
    g, o = both
    rg = g.run_correction(4, sc.selected, sc.poses, sc.cov, [], mode=abi.MODE_JOINT, stf_window=3)
    ro = o.run_correction(4, sc.selected, sc.poses, sc.cov, [], mode=abi.MODE_JOINT, stf_window=3)
    assert rg["status"] == ro["status"] == "ok"
    cg, co = g.get_correspondences(), o.get_correspondences()
    for a, b in zip(cg, co):
        assert np.array_equal(a, b)                       # matched at bit-identical poses: identical index sets
    assert rg["summary"].num_matches == ro["summary"].num_matches > 1000
    assert rg["summary"].termination == ro["summary"].termination
    assert rg["summary"].iterations == ro["summary"].iterations
    d = rg["poses"].astype(np.float64) - ro["poses"]
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    assert np.abs(d[:, :2]).max() < 1e-4 and np.abs(d[:, 2]).max() < 1e-4
    assert abs(rg["summary"].final_cost - ro["summary"].final_cost) <= 1e-6 * ro["summary"].final_cost


def test_replaying_a_recorded_session(both, sc, tmp_path):
    """Write a two-correction recording with the product's writer, read it back, replay it on both backends."""
    from hitl_slam_amd import io as HIO
    g, o = both
    sc2 = Scenario(first_piece=(10.5, 17.5))
    entries = [dict(type=4, undone=0, points=sc.selected.reshape(4, 2)), dict(type=1, undone=0, points=sc.selected.reshape(4, 2)[:2]),
               dict(type=4, undone=1, points=sc2.selected.reshape(4, 2))]
    path = str(tmp_path / "rec.log")
    HIO.save_input_log(path, entries)
    log = HIO.load_input_log(path)
    assert len(log) == 3
    pg, cg, gg, sg = g.replay_log(log, sc.poses, sc.cov)
    po, co, go, so = o.replay_log(log, sc.poses, sc.cov)
    assert sg == so and sg[0] == "ok" and sg[1] == "unsupported"
    assert len(gg) == len(go) == sg.count("ok")
    assert np.array_equal(cg.view(np.uint32), co.view(np.uint32))
    d = pg.astype(np.float64) - po
    d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
    assert np.abs(d[:, :2]).max() < 2e-4 and np.abs(d[:, 2]).max() < 2e-4


def test_repeated_calls_do_not_grow_device_memory(both, sc):
    """Every entry point allocates its scratch per call and must give it back: 40 whole cycles + covariances later the
    device has as much free memory as after the first one."""
    from hitl_slam_amd.api import device_memory
    g, _ = both

    def once():
        r = g.run_correction(4, sc.selected, sc.poses, sc.cov, [], mode=abi.MODE_JOINT, stf_window=2)
        g.pose_covariances(abi.MODE_JOINT, r["poses"])
        g.transform_scans(r["poses"])
    once()
    free0, total = device_memory()
    for _ in range(40):
        once()
    free1, _ = device_memory()
    assert total > 0 and free0 - free1 < 32 * 1024 * 1024, (free0, free1)
