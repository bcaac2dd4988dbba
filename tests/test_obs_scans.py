"""The O(N*M) observation scans either side of the solve: hand-made
known answers for the oracle's restatement, and bit-exact GPU-vs-oracle parity."""
import math

import numpy as np
import pytest

from conftest import Problem

F = np.float32


def _tiny(solver):
    # two poses, three observations each
    off = np.array([0, 3, 6], np.int64)
    px = np.array([1, 2, 3, 1, 1, 1], F)
    py = np.array([0, 0, 0, -1, 0, 1], F)
    solver.set_map(off, px, py, -np.ones(6, F), np.zeros(6, F))
    poses = np.array([[0, 0, 0], [10, 5, math.pi / 2]], F)
    return poses


def test_oracle_known_answers():
    from oracle import hitl_oracle as O
    o = O.OracleSolver()
    poses = _tiny(o)
    wx, wy = o.transform_scans(poses)
    assert np.allclose(wx[:3], [1, 2, 3]) and np.allclose(wy[:3], [0, 0, 0])
    # pose 1 heads along +y: robot (1, -1) -> world (10 + 1, 5 + 1), robot (1, 1) -> (10 - 1, 5 + 1)
    assert np.allclose(wx[3:], [11, 10, 9], atol=1e-6) and np.allclose(wy[3:], [6, 6, 6], atol=1e-6)
    seen = o.verify_points(poses, [[2.03, 0.0], [2.06, 0.0], [9.0, 6.04], [50, 50]], 0.05)
    assert seen.tolist() == [1, 0, 1, 0]                      # strictly within 5 cm
    # selection 0: segment from (0.5,0) to (2.5,0) -> world points (1,0),(2,0) inside, (3,0) is 0.5 past the end
    # selection 1: segment from (9,5.98) to (11,5.98) -> all three points of pose 1 are 2 cm away
    counts, mask = o.observation_sets(poses, [0.5, 0, 2.5, 0, 9, 5.98, 11, 5.98], 0.03)
    assert counts.tolist() == [[2, 0], [0, 3]] and mask.tolist() == [1, 1, 0, 2, 2, 2]
    # the caps of the pill are round: a point 2 cm beyond the end along the line is inside, 4 cm is not
    counts, _ = o.observation_sets(poses, [0.0, 0, 0.98, 0, 0, 0, 0.96, 0], 0.03)
    assert counts[0].tolist() == [1, 0]


@pytest.mark.gpu
def test_gpu_scans_bit_exact_vs_oracle():
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    prob = Problem(400, 180, ppf=200, groups=0)
    g, o = prob.load(H.Solver()), prob.load(O.OracleSolver())
    poses = prob.init
    wg, wo = g.transform_scans(poses), o.transform_scans(poses)
    assert np.array_equal(wg[0], wo[0]) and np.array_equal(wg[1], wo[1])
    # the same clouds into page-locked memory of the library's allocator, and left on the device (read back through torch)
    px_, py_, pinned = g.transform_scans_pinned(poses)
    assert np.array_equal(px_, wo[0]) and np.array_equal(py_, wo[1])
    g.host_free(pinned)
    dx, dy = g.transform_scans_device(poses)
    assert dx and dy and dy - dx == 4 * len(wo[0])
    import ctypes
    hip = ctypes.CDLL("libamdhip64.so")
    back = np.zeros(len(wo[0]), np.float32)
    assert hip.hipMemcpy(back.ctypes.data_as(ctypes.c_void_p), ctypes.c_void_p(dy), ctypes.c_size_t(back.nbytes), 2) == 0     # 2 = device to host
    assert np.array_equal(back, wo[1])
    wx, wy = wo
    rng = np.random.default_rng(0)
    pick = rng.integers(0, len(wx), 6)
    pts = np.stack([wx[pick] + rng.normal(0, 0.03, 6), wy[pick] + rng.normal(0, 0.03, 6)], 1).astype(F)
    pts = np.concatenate([pts, [[1e3, 1e3]]]).astype(F)
    assert np.array_equal(g.verify_points(poses, pts, 0.05), o.verify_points(poses, pts, 0.05))
    # a user-drawn pair of segments along two walls of the first floor
    y0 = wy[:200 * 180].max()
    segs = [-20.0, y0 - 0.01, -5.0, y0 + 0.01, -20.0, wy[:200 * 180].min(), 0.0, wy[:200 * 180].min()]
    cg, mg = g.observation_sets(poses, segs, 0.03)
    co, mo = o.observation_sets(poses, segs, 0.03)
    assert np.array_equal(cg, co) and np.array_equal(mg, mo)
    assert (co > 5).any()                                       # some poses really observe the selections (EMinput.cpp:309-316)
    g.close()
