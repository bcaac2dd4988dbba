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
    o.find_correspondences(prob.init, window, 1)
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


def test_solver_paths_agree_over_many_map_sizes(tmp_path):
    """Structural fuzz of the linear solve: the same LM solves (mode A: band of one pose; mode C, window 3) over a few dozen
    map sizes through (1) the fused cyclic-reduction stages, (2) the level-per-launch kernels (test hook cr_legacy) and (3) the
    submap code path on a single rank (HITL_FORCE_DIST) in both of its forms -- separators (the chain of owned super-blocks with
    its last node kept: the irregular right boundary of every stage's last chunk) and, with HITL_DIST_COLUMNS, interface columns
    (multi-right-hand-side stages, interface system of one rank).  All four
    are exact solves of the same systems: iteration counts, terminations and costs must agree for every size (a chunk-boundary
    bug of the fused stages showed up as FAILURE on one size in 32)."""
    import json
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    res = {}
    # round 5: "fused" = the default plan (two-way sequential first stage on chains of >= 4096 super-blocks, the last chunk stage + the
    # top + their back-substitutions in one launch); "seq5" forces the sequential stage with chunks of 5 on EVERY size (short, odd and
    # single-node last chunks at both band widths); "plain" = neither (the stage kernels of rounds 2-4, one launch each)
    for tag, extra in (("fused", {}), ("legacy", {"HITL_TEST_SWITCHES": "cr_legacy=1"}), ("dist", {"HITL_FORCE_DIST": "1"}),
                       ("distcols", {"HITL_FORCE_DIST": "1", "HITL_DIST_COLUMNS": "1"}), ("seq5", {"HITL_TEST_SWITCHES": "cr_seq_m=5"}),
                       ("plain", {"HITL_TEST_SWITCHES": "cr_seq=0,cr_tail=0"})):
        out = str(tmp_path / (tag + ".jsonl"))
        env = dict(os.environ, HITL_SWEEP_QUICK="1", **extra)
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "cr_size_sweep.py"), out], env=env, cwd=root, capture_output=True,
                           text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        res[tag] = [json.loads(l) for l in open(out)]
    assert len(res["fused"]) == len(res["legacy"]) == len(res["dist"]) == len(res["distcols"]) >= 30
    for a, b, c, d in zip(res["fused"], res["legacy"], res["dist"], res["distcols"]):
        assert "err" not in a and "err" not in b and "err" not in c and "err" not in d, (a, b, c, d)
        assert a["n"] == b["n"] == c["n"] == d["n"] and a["term"] != 2
        assert a["it"] == b["it"] == c["it"] == d["it"] and a["term"] == b["term"] == c["term"] == d["term"], (a, b, c, d)
        assert abs(a["cost"] - b["cost"]) <= 1e-9 * max(abs(b["cost"]), 1e-12) and abs(c["cost"] - b["cost"]) <= 1e-8 * max(abs(b["cost"]), 1e-12)
        assert abs(d["cost"] - b["cost"]) <= 1e-8 * max(abs(b["cost"]), 1e-12)
    for tag in ("seq5", "plain"):
        assert len(res[tag]) == len(res["legacy"])
        for e, b in zip(res[tag], res["legacy"]):
            assert "err" not in e and e["n"] == b["n"] and e["it"] == b["it"] and e["term"] == b["term"], (tag, e, b)
            assert abs(e["cost"] - b["cost"]) <= 1e-9 * max(abs(b["cost"]), 1e-12), (tag, e, b)


def test_solves_are_reproducible_run_to_run(tmp_path):
    """No atomics, fixed summation orders, barriers where LDS slots are re-used: the same solves twice give the same bits (a
    workgroup-level race in the cyclic reduction -- a rhs slot overwritten while a store pass still read it -- once showed up only
    as iteration counts that changed from run to run on a few map sizes)."""
    import os
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for k in range(2):
        out = str(tmp_path / ("run%d.jsonl" % k))
        env = dict(os.environ, HITL_SWEEP_QUICK="1")
        r = subprocess.run([sys.executable, os.path.join(root, "tools", "cr_size_sweep.py"), out], env=env, cwd=root, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, r.stdout[-1000:] + r.stderr[-3000:]
        outs.append(open(out).read())
    assert outs[0] == outs[1] and outs[0].count("\n") >= 30
