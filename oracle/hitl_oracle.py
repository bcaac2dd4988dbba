"""ctypes wrapper of the CPU oracle (oracle/hitl_oracle.c).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg.  The product package (hitl_slam_amd) never imports this module.
PARITY UNPINNED -- see hitl_oracle.h.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from hitl_slam_amd import abi
from hitl_slam_amd.api import SolverBase, _as, _ptr, _f32p, _f64p

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "_build", "libhitl_oracle.so")
_lib = None


def build(force=False):
    srcs = [os.path.join(_HERE, "hitl_oracle.c"), os.path.join(_HERE, "hitl_oracle.h"), os.path.join(_HERE, "..", "include", "hitl_gpu.h")]
    if force or not os.path.exists(LIB_PATH) or os.path.getmtime(LIB_PATH) < max(os.path.getmtime(f) for f in srcs):
        subprocess.check_call(["make", "-C", _HERE, "-s"] + (["-B"] if force else []))
    return LIB_PATH


def lib():
    """The checker build; with HITL_ORACLE_NATIVE=1 in the environment (bench.py's cpu_baseline leg only) the
    `make native` build for the host it runs on (-O3 -march=native, vectorised match loop)."""
    global _lib
    if _lib is None:
        if os.environ.get("HITL_ORACLE_NATIVE") == "1":
            path = os.path.join(_HERE, "_build", "libhitl_oracle_native.so")
            subprocess.check_call(["make", "-C", _HERE, "-s", "-B", "native"])   # always for THIS host's ISA
        else:
            build()          # no-op when the .so is newer than its source
            path = LIB_PATH
        _lib = C.CDLL(path)
        _bind_functors(_lib)
    return _lib


def _bind_functors(L):
    f3 = C.c_float * 3
    L.orc_angle_mod.restype = C.c_double
    L.orc_angle_mod.argtypes = [C.c_double]
    L.orc_odom_setup.argtypes = [_f32p, _f32p, _f32p, _f32p, _f32p]
    L.orc_odom_setup.restype = None
    for name in ("orc_odom_eval", "orc_odom_eval_jet"):
        fn = getattr(L, name)
        fn.argtypes = [_f64p, _f64p, _f32p, C.c_float, C.c_float, _f32p, _f64p, _f64p]
        fn.restype = None
    L.orc_hc_target.argtypes = [C.POINTER(abi.HumanConstraint), _f32p, _f32p]
    L.orc_hc_target.restype = None
    for name in ("orc_hc_eval", "orc_hc_eval_jet"):
        fn = getattr(L, name)
        fn.argtypes = [C.c_uint32, _f32p, _f64p, _f64p, _f64p]
        fn.restype = C.c_int
    for name in ("orc_stf_eval", "orc_stf_eval_jet"):
        fn = getattr(L, name)
        fn.argtypes = [_f64p, _f64p, C.c_int64] + [_f32p] * 8 + [C.c_float, C.c_float, _f64p, _f64p]
        fn.restype = None
    L.orc_set_num_threads.argtypes = [C.c_int]
    L.orc_set_num_threads.restype = C.c_int
    del f3


def set_num_threads(n):
    return lib().orc_set_num_threads(int(n))


# ---- functor-level helpers -------------------------------------------------------------
def odom_setup(pose_prev, pose_cur):
    pp, pc = _as(pose_prev, np.float32), _as(pose_cur, np.float32)
    A = np.zeros(4, np.float32)
    rho, rot = C.c_float(), C.c_float()
    lib().orc_odom_setup(_ptr(pp, _f32p), _ptr(pc, _f32p), _ptr(A, _f32p), C.byref(rho), C.byref(rot))
    return A, rho.value, rot.value


def odom_eval(x1, x2, A, rho, rot, sig=(0.03, 0.03, 0.01), jet=False):
    x1, x2 = _as(x1, np.float64), _as(x2, np.float64)
    A, sig = _as(A, np.float32), _as(sig, np.float32)
    r, J = np.zeros(3), np.zeros((3, 6))
    fn = lib().orc_odom_eval_jet if jet else lib().orc_odom_eval
    fn(_ptr(x1, _f64p), _ptr(x2, _f64p), _ptr(A, _f32p), rho, rot, _ptr(sig, _f32p), _ptr(r, _f64p), _ptr(J, _f64p))
    return r, J


def hc_target(row, anchor_pose):
    a = _as(anchor_pose, np.float32)
    t = np.zeros(4, np.float32)
    lib().orc_hc_target(C.byref(row), _ptr(a, _f32p), _ptr(t, _f32p))
    return t


def hc_eval(ctype, tgt, x, jet=False):
    tgt, x = _as(tgt, np.float32), _as(x, np.float64)
    r, J = np.zeros(3), np.zeros(9)
    fn = lib().orc_hc_eval_jet if jet else lib().orc_hc_eval
    n = fn(int(ctype), _ptr(tgt, _f32p), _ptr(x, _f64p), _ptr(r, _f64p), _ptr(J, _f64p))
    return r[:n], J[:3 * n].reshape(n, 3)


def stf_eval(x0, x1, p0, p1, n0, n1, std_dev=0.05, corr=1.0 / 40.0, jet=False):
    """p0,p1,n0,n1: float32 [n,2] matched points/normals in their own robot frames."""
    x0, x1 = _as(x0, np.float64), _as(x1, np.float64)
    cols = []
    for a in (p0, p1, n0, n1):
        a = _as(a, np.float32)
        cols += [np.ascontiguousarray(a[:, 0]), np.ascontiguousarray(a[:, 1])]
    r, J = np.zeros(2), np.zeros((2, 6))
    fn = lib().orc_stf_eval_jet if jet else lib().orc_stf_eval
    fn(_ptr(x0, _f64p), _ptr(x1, _f64p), len(cols[0]), *[_ptr(c, _f32p) for c in cols],
       C.c_float(std_dev), C.c_float(corr), _ptr(r, _f64p), _ptr(J, _f64p))
    return r, J


class OracleSolver(SolverBase):
    """Same surface as hitl_slam_amd.Solver, backed by the CPU oracle."""

    _names = dict(create="orc_create", destroy="orc_destroy", set_map="orc_set_map",
                  set_hc="orc_set_human_constraints", set_corr="orc_set_correspondences",
                  get_corr_counts="orc_get_correspondence_counts", get_corr="orc_get_correspondences",
                  solve="orc_solve", evaluate="orc_evaluate", linearize="orc_linearize")
    _scan_names = dict(transform="orc_transform_scans", verify="orc_verify_points", obs_sets="orc_observation_sets")
    _cycle_names = dict(fit="orc_fit_segments", order="orc_order_user_input", explicit="orc_explicit_correction",
                        backprop="orc_backprop", normalize="orc_normalize_angles")
    _cycle_handle = False
    _cov_name = "orc_pose_covariances"

    def __init__(self, options=None):
        super().__init__(lib(), options)
        L = self._lib
        L.orc_find_correspondences.argtypes = [C.c_void_p, _f32p, C.c_int64, C.c_int]
        L.orc_solve_f64.argtypes = [C.c_void_p, C.c_int32, _f32p, _f64p, C.POINTER(abi.Summary)]
        L.orc_assemble_dense.argtypes = [C.c_void_p, C.c_int32, _f32p, _f64p, _f64p, _f64p, _f64p]
        L.orc_set_use_jets.argtypes = [C.c_void_p, C.c_int]

    def find_correspondences(self, poses_xyt, window, method=1):
        """method 1 (default, like hitl_options.kd_tree_search = 1): the reference's KD-tree walk; 0: the brute-force definition."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        st = self._lib.orc_find_correspondences(self._h, _ptr(pin, _f32p), int(window), int(method))
        self._check(st, "create")

    def phase_ms(self):
        """Wall-clock of the last solve by phase: dict(evaluate, accumulate, scale, cholesky, bookkeeping) in ms."""
        out = (C.c_double * 8)()
        self._lib.orc_get_phase_ms.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        self._lib.orc_get_phase_ms(self._h, out)
        return dict(evaluate=out[0], accumulate=out[1], scale=out[2], cholesky=out[3], bookkeeping=out[4])

    def set_use_jets(self, use):
        self._lib.orc_set_use_jets(self._h, int(bool(use)))

    def solve_f64(self, mode, poses_xyt):
        pin = _as(poses_xyt, np.float32).reshape(-1)
        x = np.zeros(3 * self.n_poses)
        summ = abi.Summary()
        st = self._lib.orc_solve_f64(self._h, int(mode), _ptr(pin, _f32p), _ptr(x, _f64p), C.byref(summ))
        self._check(st, "solve", (abi.ERR_SOLVER_FAILURE,))
        return x.reshape(-1, 3), summ

    def assemble_dense(self, mode, poses_ref_xyt, x):
        pref = _as(poses_ref_xyt, np.float32).reshape(-1)
        x = _as(x, np.float64).reshape(-1)
        d = 3 * self.n_poses
        H, g, cost = np.zeros((d, d)), np.zeros(d), C.c_double()
        st = self._lib.orc_assemble_dense(self._h, int(mode), _ptr(pref, _f32p), _ptr(x, _f64p), _ptr(H, _f64p),
                                          _ptr(g, _f64p), C.byref(cost))
        self._check(st, "linearize")
        return H, g, cost.value


def backprop_f64(poses_xyt, cov9, lo, hi, correction):
    """Backprop::Run twice from the same inputs: the literal fp32 loops and the same loops in fp64.
    Returns (poses_f32 [n,3], cov9 [n,9], poses_f64 [n,3])."""
    poses = np.array(poses_xyt, np.float32).reshape(-1).copy()
    cov = np.array(cov9, np.float32).reshape(-1).copy()
    n = poses.size // 3
    cvec = _as(correction, np.float32).reshape(-1)
    x64 = np.zeros(3 * n, np.float64)
    f = lib().orc_backprop
    f.argtypes = [C.c_int64, _f32p, _f32p, C.c_int32, C.c_int32, _f32p, _f64p]
    f.restype = C.c_int
    st = f(n, _ptr(poses, _f32p), _ptr(cov, _f32p), int(lo), int(hi), _ptr(cvec, _f32p), _ptr(x64, _f64p))
    assert st == 0, st
    return poses.reshape(-1, 3), cov.reshape(-1, 9), x64.reshape(-1, 3)
