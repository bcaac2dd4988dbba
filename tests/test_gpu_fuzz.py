"""Seeded sweep of small problems through the whole path (match -> linearise -> solve -> covariances) on the GPU and on
the CPU oracle: sizes, scan lengths, floor layouts, matching windows, constraint kinds and modes the other parity
tests do not visit.  Index sets bit for bit; residuals/Jacobians 1e-9 relative; poses 1e-4 m / 1e-4 rad."""
import numpy as np
import pytest

from hitl_slam_amd import abi
from tests.conftest import Problem

pytestmark = pytest.mark.gpu

CASES = []
_rng = np.random.default_rng(20260301)
for _k in range(18):
    n = int(_rng.integers(9, 150))
    obs = int(_rng.integers(24, 200))
    ppf = int(_rng.integers(8, n + 1))
    window = int(_rng.integers(1, 5))
    kind = int(_rng.choice([0, 2, 4, 5, 6]))
    groups = int(_rng.integers(0, 4))
    CASES.append((n, obs, ppf, window, kind, groups, 20260400 + _k))


@pytest.mark.parametrize("n,obs,ppf,window,kind,groups,seed", CASES)
def test_seeded_problem(n, obs, ppf, window, kind, groups, seed):
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    prob = Problem(n, obs, seed=seed, ppf=ppf, groups=groups, kind=kind, window=window)
    g, o = H.Solver(), O.OracleSolver()
    prob.load(g)
    prob.load(o)
    g.find_correspondences(prob.init, window)
    o.find_correspondences(prob.init, window, 0)
    cg, co = g.get_correspondences(), o.get_correspondences()
    for a, b in zip(cg, co):
        assert np.array_equal(a, b)
    n_pairs = len(cg[0])
    rng = np.random.default_rng(seed)
    x = prob.init.astype(np.float64) + 0.01 * rng.standard_normal(prob.init.shape)
    modes = [abi.MODE_ODOM_HC] + ([abi.MODE_JOINT] if n_pairs else [])
    for mode in modes:
        c1, g1, r1, j1 = g.linearize(mode, prob.init, x)
        c2, g2, r2, j2 = o.linearize(mode, prob.init, x)
        assert abs(c1 - c2) <= 1e-10 * abs(c2)
        assert np.abs(g1 - g2).max() <= 1e-9 * max(np.abs(g2).max(), 1e-300)
        if mode != abi.MODE_ODOM_HC:
            assert np.abs(j1 - j2).max() <= 1e-9 * np.abs(j2).max()
        pg, sg = g.solve(mode, prob.init)
        po, so = o.solve(mode, prob.init)
        assert sg.termination == so.termination
        assert sg.iterations == so.iterations, (sg.iterations, so.iterations)
        d = pg.astype(np.float64) - po
        d[:, 2] = np.angle(np.exp(1j * d[:, 2]))
        assert np.abs(d[:, :2]).max() < 1e-4 and np.abs(d[:, 2]).max() < 1e-4
        Sg, So = g.pose_covariances(mode, po), o.pose_covariances(mode, po)
        scale = np.abs(So).max(axis=(1, 2))[:, None, None] + 1e-300
        assert (np.abs(Sg - So) / scale).max() < 1e-6
    g.close(); o.close()
