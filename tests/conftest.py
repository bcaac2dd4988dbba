import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


# The oracle is an OpenMP program.  A GPU box shows 256 CPUs but grants 16 cores (cgroup quota): 256 threads on them make
# every small oracle solve take seconds (7.5 s for 104 poses).  Use what can really run at once -- before the library loads.
if "OMP_NUM_THREADS" not in os.environ:
    from tools.probe_host import usable_cores
    os.environ["OMP_NUM_THREADS"] = str(max(1, min(usable_cores(), 16)))


def pytest_configure(config):
    config.addinivalue_line("markers", "gpu: needs a real MI355X (run with -m gpu on the GPU box)")


@pytest.fixture
def switches():
    """Test hooks of libhitlgpu.so (hitl_test_set_switch, csrc/hitl_host.h): `switches(stf_addr64=1, kd_cap=1)` sets them for
    the solvers the test creates afterwards; all are cleared when the test ends."""
    import hitl_slam_amd as H

    def setter(**kw):
        for k, v in kw.items():
            H.set_test_switch(k, v)
    yield setter
    H.set_test_switch(None)


@pytest.fixture(scope="session", autouse=True)
def _native_built():
    """Build the oracle and the synthetic-map tool (gcc, seconds).  libhitlgpu.so is built by
    __graft_entry__.build(); GPU tests fail loudly if it is missing."""
    from oracle import hitl_oracle
    from tools import hitl_synth
    hitl_oracle.build()
    hitl_synth.build()


class Problem(object):
    """A small synthetic map + constraints, shared by oracle-only and GPU parity tests."""

    def __init__(self, n_poses, obs, seed=20260102, ppf=None, groups=2, kind=4, window=3):
        from tools import hitl_synth as S
        self.p = S.params(n_poses, obs, seed=seed, poses_per_floor=ppf or max(n_poses // 2, 8))
        self.gt, self.init, self.cov = S.poses(self.p)
        self.off, self.px, self.py, self.nx, self.ny = S.scans(self.p)
        self.rows = S.constraints(self.p, self.gt, groups, kind=kind) if groups else (None, 0)
        self.window = window
        self.n = n_poses

    def load(self, solver, with_corr=True, corr=None):
        solver.set_map(self.off, self.px, self.py, self.nx, self.ny, self.cov)
        if self.rows[1]:
            solver.set_human_constraints(self.rows)
        if corr is not None:
            solver.set_correspondences(*corr)
        return solver


@pytest.fixture(scope="session")
def small_problem():
    return Problem(60, 90, ppf=30)


# the oracle's search method that matches the product's default (hitl_options.kd_tree_search = 1): the reference's own
# KD-tree walk (method 1); method 0 is the brute-force definition (kd_tree_search = 0)
REF_SEARCH = 1


def oracle_corr(prob, window=None, method=REF_SEARCH):
    from oracle import hitl_oracle as O
    o = O.OracleSolver()
    prob.load(o)
    o.find_correspondences(prob.init, prob.window if window is None else window, method)
    c = o.get_correspondences()
    o.close()
    return c
