"""Opt-in hitl_options.stf_moments: the scan-to-scan factors evaluated from 2 x 15 moments per pose pair instead of
streaming every match.  Same algebra as the default path (and as residual_functors.h:786-830), different summation
order: compared against the streaming path and against the oracle."""
import numpy as np
import pytest

import hitl_slam_amd as H
from hitl_slam_amd import abi
from tests.conftest import Problem, oracle_corr

pytestmark = pytest.mark.gpu


def _solvers(prob, corr):
    o1, o2 = H.default_options(), H.default_options()
    o2.stf_moments = 1
    a, b = H.Solver(o1), H.Solver(o2)
    prob.load(a, corr=corr)
    prob.load(b, corr=corr)
    return a, b


@pytest.mark.parametrize("n,obs,ppf,window", [(60, 90, 30, 3), (97, 64, 97, 2), (130, 48, 50, 4)])
def test_linearisation_matches_the_streaming_path(n, obs, ppf, window):
    prob = Problem(n, obs, ppf=ppf, window=window)
    corr = oracle_corr(prob)
    a, b = _solvers(prob, corr)
    rng = np.random.default_rng(n)
    x = prob.init.astype(np.float64) + 0.02 * rng.standard_normal(prob.init.shape)
    for mode in (abi.MODE_STF_ONLY, abi.MODE_JOINT):
        ca, ga, ra, ja = a.linearize(mode, prob.init, x)
        cb, gb, rb, jb = b.linearize(mode, prob.init, x)
        assert abs(ca - cb) <= 1e-9 * ca
        assert np.abs(ra - rb).max() <= 1e-8 * np.abs(ra).max()
        assert np.abs(ja - jb).max() <= 1e-8 * np.abs(ja).max()
        assert np.abs(ga - gb).max() <= 1e-8 * np.abs(ga).max()
    a.close(); b.close()


@pytest.mark.parametrize("n,obs,ppf,window", [(60, 90, 30, 3), (130, 48, 50, 4)])
def test_solve_matches_the_streaming_path_and_the_oracle(n, obs, ppf, window):
    from oracle import hitl_oracle as O
    prob = Problem(n, obs, ppf=ppf, window=window)
    corr = oracle_corr(prob)
    a, b = _solvers(prob, corr)
    o = O.OracleSolver()
    prob.load(o, corr=corr)
    pa, sa = a.solve(abi.MODE_JOINT, prob.init)
    pb, sb = b.solve(abi.MODE_JOINT, prob.init)
    po, so = o.solve(abi.MODE_JOINT, prob.init)
    assert sa.iterations == sb.iterations == so.iterations and sa.termination == sb.termination == so.termination
    assert abs(sa.final_cost - sb.final_cost) <= 1e-8 * sa.final_cost
    assert np.abs(pa - pb).max() <= 1e-6
    assert np.abs(pb[:, :2] - po[:, :2]).max() < 1e-4 and np.abs(pb[:, 2] - po[:, 2]).max() < 1e-4
    # a new correspondence set replaces the moments
    b.find_correspondences(pb, window)
    a.find_correspondences(pa, window)
    pa2, sa2 = a.solve(abi.MODE_JOINT, pa)
    pb2, sb2 = b.solve(abi.MODE_JOINT, pb)
    assert sa2.iterations == sb2.iterations and np.abs(pa2 - pb2).max() <= 1e-5
    a.close(); b.close(); o.close()


def test_create_solve_destroy_gives_all_device_memory_back():
    from hitl_slam_amd.api import device_memory
    prob = Problem(60, 90, ppf=30, window=3)
    corr = oracle_corr(prob)

    def once(moments):
        o = H.default_options()
        o.stf_moments = moments
        s = H.Solver(o)
        prob.load(s, corr=corr)
        s.find_correspondences(prob.init, 3)
        s.solve(abi.MODE_JOINT, prob.init)
        s.pose_covariances(abi.MODE_JOINT, prob.init)
        s.close()
    once(0); once(1)
    free0, _ = device_memory()
    for k in range(12):
        once(k & 1)
    free1, _ = device_memory()
    assert free0 - free1 < 16 * 1024 * 1024, (free0, free1)
