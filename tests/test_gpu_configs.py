"""BASELINE.json configs 1 and 3 on the GPU, through the real file loader: a Figure-8 drive
(stand-in written in the reference's .stfs.covars format, tests/figure8.py) without constraints,
and with 5 loop-closure + 3 colinear human-constraint groups.  The all-pairs correspondence
search (the reference's, JointOptimization.cpp:561-642) finds the lap-to-lap revisits, so mode C
carries off-band pose pairs and the PCG really iterates."""
import numpy as np
import pytest

import figure8
from hitl_slam_amd import abi
from hitl_slam_amd import io as hio
from hitl_slam_amd.api import hc_rows

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def fig8(tmp_path_factory):
    d = figure8.make(160, 90)
    path = str(tmp_path_factory.mktemp("fig8") / "figure8_standin.bag.stfs.covars")
    hio.save_stfs_covars(path, "AMRL", 1455638506.25, d["init"], d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
    g = hio.load_stfs_covars(path, fix_normals=True)
    return g, d


def _pair(g):
    import hitl_slam_amd as H
    from oracle import hitl_oracle as O
    out = []
    for s in (H.Solver(), O.OracleSolver()):
        s.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
        out.append(s)
    return out


def _wrap(d):
    d = d.copy()
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))
    return d


def test_config1_figure8_no_constraints(fig8):
    g, _ = fig8
    gpu, orc = _pair(g)
    pg, sg = gpu.solve(abi.MODE_ODOM_HC, g["poses"])
    po, so = orc.solve(abi.MODE_ODOM_HC, g["poses"])
    assert sg.termination == so.termination == abi.CONVERGENCE and sg.iterations == so.iterations
    assert np.abs(_wrap(pg - g["poses"])).max() <= 2e-6 and np.abs(pg - po).max() <= 1e-6


@pytest.mark.parametrize("mode", [abi.MODE_ODOM_HC, abi.MODE_JOINT])
def test_config3_figure8_with_loop_and_colinear_constraints(fig8, mode):
    g, d = fig8
    rows = hc_rows(figure8.constraints(d["gt"], 5, 3))
    assert rows[1] == 8 * 9
    gpu, orc = _pair(g)
    for s in (gpu, orc):
        s.set_human_constraints(rows)
    if mode == abi.MODE_JOINT:
        gpu.find_correspondences(g["poses"], 0)            # all pairs, like the reference
        orc.find_correspondences(g["poses"], 0, 0)
        cg, co = gpu.get_correspondences(), orc.get_correspondences()
        for a, b in zip(cg, co):
            assert np.array_equal(a, b)
        assert np.abs(cg[0] - cg[1]).max() > 40           # lap-to-lap revisits were matched
    pg, sg = gpu.solve(mode, g["poses"])
    po, so = orc.solve(mode, g["poses"])
    assert sg.termination == so.termination and sg.iterations == so.iterations
    assert sg.num_hc_residuals == so.num_hc_residuals == 5 * 9 * 3 + 3 * 9 * 2
    assert abs(sg.final_cost - so.final_cost) <= 1e-8 * so.final_cost
    assert np.abs(pg[:, :2] - po[:, :2]).max() <= 1e-4 and np.abs(pg[:, 2] - po[:, 2]).max() <= 1e-4
    if mode == abi.MODE_JOINT:
        assert sg.num_offband_pairs > 0 and sg.linear_iterations > sg.iterations
    assert sg.final_cost < sg.initial_cost
