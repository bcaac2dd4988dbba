"""The occupancy grids in front of the source-major tree search (round 6, hitl_slam_amd/csrc/hitl_match.hip: k_kd_grid + the bit test
in k_match_kd): a query with no scan point of the target within the point-match threshold skips its KD walk.  The reference's walk
(kdtree.cpp:141-197) only ever returns a node inside that gate (:147), so the skip may not change ONE correspondence: held here to
the walk-everything kernel of rounds 3-5 (test hook kd_grid = 0) bit for bit -- pairs, match offsets, both index lists -- on maps
that revisit places, at BASELINE configs[2]'s size, over thresholds from 2 cm to 5 m (cells from the finest to coarser than the
room), on scans far from their pose's origin (cells coarsened until the grid fits its 32 768 bits), on scans with NaN points and
scans with no points, and to the oracle's walk.
The same kernel takes G = 4 (or 2) targets per workgroup at a time (k_match_kd_spec: walks first, the per-point cap of
JointOptimization.cpp:596-599 applied afterwards in target order); test hook kd_spec = 0 is the one-target-at-a-time order of rounds
3-5, and every comparison below is made against it as well -- including a cap of 1 and 2 matches per point (the cap closes points
inside a group) and several tiles of source poses."""
import numpy as np
import pytest

import figure8
from conftest import Problem

pytestmark = pytest.mark.gpu


def _search(H, switches, grid, opt, load, init, window=0, spec=4, **more):
    switches(kd_grid=grid, kd_spec=spec, **more)
    s = H.Solver(opt)
    load(s)
    s.find_correspondences(init, window)
    c = s.get_correspondences()
    s.close()
    return c


def _same(a, b):
    assert len(a) == len(b)
    for x, y in zip(a, b):
        assert np.array_equal(x, y)


def _fig8_loader(d):
    def load(s):
        s.set_map(d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
    return load


@pytest.mark.parametrize("thr", [0.02, 0.15, 0.6, 5.0])
def test_grid_skips_change_no_correspondence_figure8(switches, thr):
    import hitl_slam_amd as H
    d = figure8.make(300, 180, seed=11, laps=3)
    opt = H.default_options()
    opt.point_match_threshold = thr
    a = _search(H, switches, 1, opt, _fig8_loader(d), d["init"])
    b = _search(H, switches, 0, opt, _fig8_loader(d), d["init"], spec=0)
    _same(a, b)
    _same(_search(H, switches, 0, opt, _fig8_loader(d), d["init"], spec=4), b)
    _same(_search(H, switches, 1, opt, _fig8_loader(d), d["init"], spec=2), b)
    print("figure-8 300 x 180 all pairs, threshold %.2f: %d pairs, %d matches" % (thr, len(a[0]), len(a[3])))
    assert len(a[3]) > 0


@pytest.mark.parametrize("cap,skip", [(1, 1), (2, 1), (3, 2), (1000, 1)])
def test_groups_of_targets_commit_in_the_sequential_order(switches, cap, skip):
    """The per-point cap at 1, 2, 3 (every group of four targets closes points half-way) and out of reach; every second reading
    skipped; one tile and several (cand_tile_bytes: the table budget of a tile): G = 4 and G = 2 against one target at a time."""
    import hitl_slam_amd as H
    d = figure8.make(230, 150, seed=3, laps=3)
    opt = H.default_options()
    opt.max_correspondences_per_point = cap
    opt.num_skip_readings = skip
    ref = _search(H, switches, 0, opt, _fig8_loader(d), d["init"], spec=0)
    assert len(ref[3]) > 0
    for spec, more in ((4, {}), (2, {}), (4, {"cand_tile_bytes": 229 * 150 * 2 * 37})):
        _same(_search(H, switches, 1, opt, _fig8_loader(d), d["init"], spec=spec, **more), ref)


def test_grid_skips_at_configs2_size_and_against_the_oracle(switches):
    """1 000 x 360 (bench.py --workload c3's map): grid against no grid; 400 x 120: against the oracle's walk."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d = figure8.make(1000, 360, seed=7, laps=4)
    opt = H.default_options()
    a = _search(H, switches, 1, opt, _fig8_loader(d), d["init"])
    b = _search(H, switches, 0, opt, _fig8_loader(d), d["init"], spec=0)
    _same(a, b)
    assert len(a[3]) > 1000000
    d = figure8.make(400, 120, seed=7, laps=2)
    a = _search(H, switches, 1, opt, _fig8_loader(d), d["init"])
    o = O.OracleSolver()
    _fig8_loader(d)(o)
    o.find_correspondences(d["init"], 0, 1)
    _same(a, o.get_correspondences())
    o.close()


def test_grid_far_scans_nan_points_and_empty_scans(switches):
    """Scans whose points lie hundreds of metres from the pose (a grid of 0.19 m cells would need millions of bits: coarser cells),
    a few NaN points, two poses without any point: the same sets with and without the grid."""
    import hitl_slam_amd as H
    prob = Problem(96, 64, ppf=96, groups=0)
    rng = np.random.default_rng(5)
    off = np.asarray(prob.off).copy()
    px, py = np.asarray(prob.px, np.float32).copy(), np.asarray(prob.py, np.float32).copy()
    nx, ny = np.asarray(prob.nx, np.float32).copy(), np.asarray(prob.ny, np.float32).copy()
    # every fourth scan: one outlier return 300-900 m away (a window, a mirror): the scan's bounding box explodes
    for i in range(0, prob.n, 4):
        k = off[i] + int(rng.integers(0, off[i + 1] - off[i]))
        px[k] = np.float32(rng.uniform(300, 900)); py[k] = np.float32(-rng.uniform(300, 900))
    # NaN returns
    for k in rng.integers(0, len(px), 40):
        px[k] = np.float32(np.nan)
    # an infinite return in two scans (no grid can be laid over those: all their queries walk)
    for i in (21, 70):
        px[off[i] + 3] = np.float32(np.inf)
    # two scans without points
    keep = np.ones(len(px), bool)
    cnt = np.diff(off).copy()
    for i in (10, 57):
        keep[off[i]:off[i + 1]] = False
        cnt[i] = 0
    off2 = np.concatenate([[0], np.cumsum(cnt)]).astype(off.dtype)
    px, py, nx, ny = px[keep], py[keep], nx[keep], ny[keep]

    def load(s):
        s.set_map(off2, px, py, nx, ny, prob.cov)
    opt = H.default_options()
    a = _search(H, switches, 1, opt, load, prob.init)
    b = _search(H, switches, 0, opt, load, prob.init, spec=0)
    _same(a, b)
    _same(_search(H, switches, 1, opt, load, prob.init, spec=2), b)
    assert len(a[3]) > 0
    # the banded (target-major) search does not use the grids: untouched by the switch
    a = _search(H, switches, 1, opt, load, prob.init, window=3)
    b = _search(H, switches, 0, opt, load, prob.init, window=3)
    _same(a, b)


@pytest.mark.parametrize("n,m", [(120, 600), (60, 1200)])
def test_long_scans_take_fewer_targets_at_a_time(switches, n, m):
    """Scans of 600 points: two targets per workgroup at a time (four trees would not leave room for two workgroups per CU); 1 200
    points: one (the rounds-3-5 order) -- picked by the launcher from the LDS budget, not by a switch; against the forced
    one-at-a-time kernel without grids and against the oracle's walk."""
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    d = figure8.make(n, m, seed=5, laps=2)
    opt = H.default_options()
    a = _search(H, switches, 1, opt, _fig8_loader(d), d["init"])
    _same(a, _search(H, switches, 0, opt, _fig8_loader(d), d["init"], spec=0))
    o = O.OracleSolver()
    _fig8_loader(d)(o)
    o.find_correspondences(d["init"], 0, 1)
    _same(a, o.get_correspondences())
    o.close()
    assert len(a[3]) > 0


def test_repeated_searches_start_the_long_workgroups_first(switches):
    """From the second search of a map on, the source-major kernel starts the workgroups of the poses that took longest last time
    first (its end is its slowest workgroup's): the mapping of workgroups to poses changes, the result may not -- three searches in a
    row (moved poses in between), against a solver that never reorders (test hook kd_order = 0).  A pose that took much longer than
    the average is then served by several workgroups, each with every parts-th point (exact: the cap counts per point): with the
    automatic choice, with EVERY pose in two and in three parts (kd_split = 2, 3), and without splitting (kd_split = 0)."""
    import hitl_slam_amd as H
    d = figure8.make(260, 140, seed=9, laps=3)
    opt = H.default_options()
    res = {}
    for order, split in ((1, 1), (0, 1), (1, 3), (1, 2), (1, 0)):
        switches(kd_order=order, kd_split=split)
        s = H.Solver(opt)
        _fig8_loader(d)(s)
        out = []
        for k in range(3):
            poses = d["init"].copy()
            poses[::3, 0] += np.float32(0.02 * k)                  # (every third pose moves: relative poses change)
            poses[1::5, 2] += np.float32(0.01 * k)
            s.find_correspondences(poses, 0)
            out.append(s.get_correspondences())
        s.close()
        res[(order, split)] = out
    for key in ((1, 1), (1, 3), (1, 2), (1, 0)):
        for a, b in zip(res[key], res[(0, 1)]):
            _same(a, b)
    assert not np.array_equal(res[(1, 1)][0][3], res[(1, 1)][2][3])          # (the moved poses did change the sets: the three searches are three problems)
