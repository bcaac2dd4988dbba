"""Host-side mirror of the reference's JointOpt interface over the C ABI (include/hitl_gpu.h).

The reference seam is the JointOpt object (JointOptimization.h:49-89): public members
poses_, robot_frame_point_clouds_, robot_frame_normal_clouds_, covariances_,
human_constraints_ in, Run(), poses_ out.  `Solver` keeps those names' meaning:

    s = Solver()                                  # JointOpt::JointOpt()
    s.set_map(obs_offset, px, py, nx, ny, cov9)   # robot_frame_*_clouds_, covariances_
    s.set_human_constraints(rows)                 # human_constraints_
    s.find_correspondences(poses, window)         # FindSTFCorrespondences (:561-642)
    poses, summary = s.solve(MODE_JOINT, poses)   # the ceres::Solve call sites (:1093, :1208)

There is no CPU fallback: constructing a Solver raises if libhitlgpu.so is missing or no
HIP device is usable.
"""
import ctypes as C
import sys
import os

import numpy as np

from . import abi

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("HITL_LIBHITLGPU") or os.path.join(_HERE, "csrc", "libhitlgpu.so")   # (the variable: same-box A/B runs against another build)

_f32p = C.POINTER(C.c_float)
_f64p = C.POINTER(C.c_double)
_i32p = C.POINTER(C.c_int32)
_i64p = C.POINTER(C.c_int64)


class HitlError(RuntimeError):
    def __init__(self, status, where, detail=""):
        self.status = status
        super().__init__("%s failed with status %d%s" % (where, status, (": " + detail) if detail else ""))


def _ptr(a, typ):
    if a is None:
        return None
    return a.ctypes.data_as(typ)


def _as(a, dtype):
    if a is None:
        return None
    return np.ascontiguousarray(a, dtype=dtype)


def hc_rows(rows):
    """Build a ctypes array of hitl_human_constraint_t from an iterable of
    (type, constrained_pose_id, anchor_pose_id, d_par, d_perp, d_angle, rel_penalty_dir)
    tuples or a structured/2-D numpy array with those 7 columns."""
    rows = list(rows) if not isinstance(rows, np.ndarray) else rows
    n = len(rows)
    arr = (abi.HumanConstraint * max(n, 1))()
    for k in range(n):
        r = rows[k]
        arr[k].type = int(r[0])
        arr[k].constrained_pose_id = int(r[1])
        arr[k].anchor_pose_id = int(r[2])
        arr[k].delta_parallel = float(r[3])
        arr[k].delta_perpendicular = float(r[4])
        arr[k].delta_angle = float(r[5])
        arr[k].relative_penalty_dir = float(r[6])
    return arr, n


class SolverBase(object):
    """Shared ctypes plumbing; subclasses bind a library and a function-name table."""

    _names = {}

    def __init__(self, lib, options=None):
        self._lib = lib
        self._h = C.c_void_p()
        self.options = abi.copy_options(options) if options is not None else abi.default_options()
        self.n_poses = 0
        self._bind()
        st = self._fn("create")(C.byref(self.options), C.byref(self._h))
        if st != abi.OK:
            self._h = C.c_void_p()
            raise HitlError(st, self._names["create"], self._detail())

    # -- binding ---------------------------------------------------------------------------
    def _fn(self, key):
        return getattr(self._lib, self._names[key])

    def _detail(self):
        return ""

    def _bind(self):
        L = self._fn
        vp = C.c_void_p
        L("create").argtypes = [C.POINTER(abi.Options), C.POINTER(vp)]
        L("create").restype = C.c_int
        L("destroy").argtypes = [vp]
        L("destroy").restype = None
        L("set_map").argtypes = [vp, C.c_int64, _i64p, _f32p, _f32p, _f32p, _f32p, _f32p]
        L("set_hc").argtypes = [vp, C.c_int64, C.POINTER(abi.HumanConstraint)]
        L("set_corr").argtypes = [vp, C.c_int64, _i32p, _i32p, _i64p, _i32p, _i32p]
        L("get_corr_counts").argtypes = [vp, _i64p, _i64p]
        L("get_corr").argtypes = [vp, _i32p, _i32p, _i64p, _i32p, _i32p]
        L("solve").argtypes = [vp, C.c_int32, _f32p, _f32p, C.POINTER(abi.Summary)]
        L("evaluate").argtypes = [vp, C.c_int32, _f32p, _f64p, _f64p]
        L("linearize").argtypes = [vp, C.c_int32, _f32p, _f64p, _f64p, _f64p, _f64p, _f64p]
        for k in ("set_map", "set_hc", "set_corr", "get_corr_counts", "get_corr", "solve", "evaluate", "linearize"):
            L(k).restype = C.c_int

    def _check(self, st, key, allow=()):
        if st != abi.OK and st not in allow:
            raise HitlError(st, self._names[key], self._detail())
        return st

    # -- lifetime --------------------------------------------------------------------------
    def close(self):
        if getattr(self, "_h", None) is not None and self._h.value:
            self._drop_world_buf()                                  # the page-locked display clouds of world_clouds()
            self._fn("destroy")(self._h)
            self._h = C.c_void_p()

    def _drop_world_buf(self):
        """Frees the page-locked staging buffers of world_clouds() (set_map: they are sized by the map; close)."""
        buf = getattr(self, "_world_buf", None)
        if buf is not None and hasattr(self, "host_free"):
            self.host_free(buf)
        self._world_buf, self._world_buf_len = None, 0

    def __del__(self):
        try:
            if sys.is_finalizing():      # interpreter shutdown: the HIP runtime may already be gone; the OS reclaims the rest
                return
            self.close()
        except Exception:
            pass

    def __enter__(self):
        return self

    def __exit__(self, *a):
        self.close()

    # -- inputs ----------------------------------------------------------------------------
    def set_map(self, obs_offset, px, py, nx, ny, cov9=None):
        obs_offset = _as(obs_offset, np.int64)
        px, py, nx, ny = (_as(a, np.float32) for a in (px, py, nx, ny))
        cov9 = _as(cov9, np.float32)
        n = len(obs_offset) - 1
        self._check(self._fn("set_map")(self._h, n, _ptr(obs_offset, _i64p), _ptr(px, _f32p), _ptr(py, _f32p),
                                        _ptr(nx, _f32p), _ptr(ny, _f32p), _ptr(cov9, _f32p)), "set_map")
        self.n_poses = n
        self.obs_offset = obs_offset
        self._drop_world_buf()          # sized by the previous map's observation count (a larger map overflowed them)

    def set_human_constraints(self, rows):
        if isinstance(rows, np.ndarray) and rows.dtype == abi.HC_DTYPE:
            rows = np.ascontiguousarray(rows)
            self._hc_keepalive = rows
            arr, n = C.cast(rows.ctypes.data_as(C.c_void_p), C.POINTER(abi.HumanConstraint)), int(rows.size)
        elif isinstance(rows, tuple) and len(rows) == 2 and isinstance(rows[1], int):
            arr, n = rows
        else:
            arr, n = hc_rows(rows)
        self._check(self._fn("set_hc")(self._h, n, arr), "set_hc")

    def set_correspondences(self, pair_src, pair_tgt, match_offset, src_idx, tgt_idx):
        pair_src, pair_tgt = _as(pair_src, np.int32), _as(pair_tgt, np.int32)
        match_offset = _as(match_offset, np.int64)
        src_idx, tgt_idx = _as(src_idx, np.int32), _as(tgt_idx, np.int32)
        self._check(self._fn("set_corr")(self._h, len(pair_src), _ptr(pair_src, _i32p), _ptr(pair_tgt, _i32p),
                                         _ptr(match_offset, _i64p), _ptr(src_idx, _i32p), _ptr(tgt_idx, _i32p)),
                    "set_corr")

    def correspondence_counts(self):
        """(directed pose pairs, point matches) held by THIS solver (local counts for a submap rank)."""
        P, Cn = C.c_int64(), C.c_int64()
        self._check(self._fn("get_corr_counts")(self._h, C.byref(P), C.byref(Cn)), "get_corr_counts")
        return P.value, Cn.value

    def get_correspondences(self):
        P, Cn = C.c_int64(), C.c_int64()
        self._check(self._fn("get_corr_counts")(self._h, C.byref(P), C.byref(Cn)), "get_corr_counts")
        P, Cn = P.value, Cn.value
        ps = np.zeros(max(P, 1), np.int32)
        pt = np.zeros(max(P, 1), np.int32)
        mo = np.zeros(P + 1, np.int64)
        i0 = np.zeros(max(Cn, 1), np.int32)
        i1 = np.zeros(max(Cn, 1), np.int32)
        self._check(self._fn("get_corr")(self._h, _ptr(ps, _i32p), _ptr(pt, _i32p), _ptr(mo, _i64p),
                                         _ptr(i0, _i32p), _ptr(i1, _i32p)), "get_corr")
        return ps[:P], pt[:P], mo, i0[:Cn], i1[:Cn]

    # -- observation scans either side of the solve -------------------
    def _scan_fn(self, key):
        return getattr(self._lib, self._scan_names[key])

    def transform_scans(self, poses_xyt):
        """== HitLSLAM::transformPointCloudsToWorldFrame: (world_x, world_y), one value per observation."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        n_obs = int(self.obs_offset[-1])
        wx, wy = np.zeros(max(n_obs, 1), np.float32), np.zeros(max(n_obs, 1), np.float32)
        f = self._scan_fn("transform")
        f.argtypes = [C.c_void_p, _f32p, _f32p, _f32p]
        f.restype = C.c_int
        self._check(f(self._h, _ptr(pin, _f32p), _ptr(wx, _f32p), _ptr(wy, _f32p)), "solve")
        return wx[:n_obs], wy[:n_obs]

    def verify_points(self, poses_xyt, points_xy, threshold=0.05):
        """== the scan of HitLSLAM::verifyUserInput: flags[q] = some observation within threshold of point q."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        pts = _as(points_xy, np.float32).reshape(-1, 2)
        seen = np.zeros(max(len(pts), 1), np.int32)
        f = self._scan_fn("verify")
        f.argtypes = [C.c_void_p, _f32p, C.c_int32, _f32p, C.c_float, _i32p]
        f.restype = C.c_int
        self._check(f(self._h, _ptr(pin, _f32p), len(pts), _ptr(pts, _f32p), C.c_float(threshold), _ptr(seen, _i32p)), "solve")
        return seen[:len(pts)]

    def observation_sets(self, poses_xyt, seg_xy8, threshold=0.03, want_mask=True):
        """== EMInput::EstablishObservationSets: (counts int32 [n,2], mask uint8 [n_obs] or None)."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        seg = _as(seg_xy8, np.float32).reshape(-1)
        assert seg.size == 8
        counts = np.zeros((len(self.obs_offset) - 1, 2), np.int32)      # resident poses (== n_poses on a single GPU)
        n_obs = int(self.obs_offset[-1])
        mask = np.zeros(max(n_obs, 1), np.uint8) if want_mask else None
        f = self._scan_fn("obs_sets")
        f.argtypes = [C.c_void_p, _f32p, _f32p, C.c_double, _i32p, C.POINTER(C.c_uint8)]
        f.restype = C.c_int
        self._check(f(self._h, _ptr(pin, _f32p), _ptr(seg, _f32p), C.c_double(threshold), _ptr(counts, _i32p),
                      _ptr(mask, C.POINTER(C.c_uint8))), "solve")
        return counts, (mask[:n_obs] if want_mask else None)

    # -- the correction cycle around the solve (HitLSLAM::Run) ------------------------------
    def _cycle_fn(self, key):
        return getattr(self._lib, self._cycle_names[key])

    def _cycle_check(self, st, key):
        if st != abi.OK:
            raise HitlError(st, self._cycle_names[key], self._detail())

    def fit_segments(self, poses_xyt, segments_xy4):
        """== EMInput::AutomaticEndpointAdjustment: (segments [k,4], inliers [k], passes [k], lm iterations [k])."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        seg = np.array(segments_xy4, np.float32).reshape(-1, 4).copy()
        k = seg.shape[0]
        inl, pas, its = np.zeros(k, np.int32), np.zeros(k, np.int32), np.zeros(k, np.int32)
        f = self._cycle_fn("fit")
        f.argtypes = [C.c_void_p, _f32p, C.c_int32, _f32p, _i32p, _i32p, _i32p]
        f.restype = C.c_int
        self._cycle_check(f(self._h, _ptr(pin, _f32p), k, _ptr(seg, _f32p), _ptr(inl, _i32p), _ptr(pas, _i32p), _ptr(its, _i32p)), "fit")
        return seg, inl, pas, its

    def order_user_input(self, counts, seg_xy8):
        """== EMInput::OrderAndFilterUserInput: dict(corrected, anchor, bounds, seg_xy8)."""
        counts = _as(counts, np.int32).reshape(-1, 2)
        n = counts.shape[0]
        seg = np.array(seg_xy8, np.float32).reshape(-1).copy()
        assert seg.size == 8
        corr, anch = np.zeros(max(n, 1), np.int32), np.zeros(max(n, 1), np.int32)
        nc, na = C.c_int32(), C.c_int32()
        bounds = np.zeros(2, np.int32)
        f = self._cycle_fn("order")
        f.argtypes = [C.c_int64, _i32p, _f32p, _i32p, C.POINTER(C.c_int32), _i32p, C.POINTER(C.c_int32), _i32p]
        f.restype = C.c_int
        self._cycle_check(f(n, _ptr(counts, _i32p), _ptr(seg, _f32p), _ptr(corr, _i32p), C.byref(nc), _ptr(anch, _i32p), C.byref(na),
                            _ptr(bounds, _i32p)), "order")
        return dict(corrected=corr[:nc.value].copy(), anchor=anch[:na.value].copy(), bounds=(int(bounds[0]), int(bounds[1])),
                    seg_xy8=seg)

    def apply_explicit_correction(self, ctype, seg_xy8, poses_xyt, corrected, anchor):
        """== AppExpCorrect::Run: (poses [n,3], correction [3], rows as abi.HC_DTYPE [n_anchor*n_corrected])."""
        poses = np.array(poses_xyt, np.float32).reshape(-1).copy()
        n = poses.size // 3
        seg = _as(seg_xy8, np.float32).reshape(-1)
        corr, anch = _as(corrected, np.int32).reshape(-1), _as(anchor, np.int32).reshape(-1)
        rows = np.zeros(max(corr.size * anch.size, 1), abi.HC_DTYPE)
        cvec = np.zeros(3, np.float32)
        f = self._cycle_fn("explicit")
        head = [C.c_void_p] if self._cycle_handle else []
        f.argtypes = head + [C.c_uint32, _f32p, C.c_int64, _f32p, C.c_int32, _i32p, C.c_int32, _i32p, _f32p, C.c_void_p]
        f.restype = C.c_int
        args = ([self._h] if self._cycle_handle else []) + [
            int(ctype), _ptr(seg, _f32p), n, _ptr(poses, _f32p), corr.size, _ptr(corr, _i32p), anch.size, _ptr(anch, _i32p),
            _ptr(cvec, _f32p), rows.ctypes.data_as(C.c_void_p)]
        self._cycle_check(f(*args), "explicit")
        return poses.reshape(-1, 3), cvec, rows[:corr.size * anch.size]

    def backprop(self, poses_xyt, cov9, lo, hi, correction):
        """== Backprop::Run: (poses [n,3], cov9 [n,9])."""
        poses = np.array(poses_xyt, np.float32).reshape(-1).copy()
        cov = np.array(cov9, np.float32).reshape(-1).copy()
        n = poses.size // 3
        assert cov.size == 9 * n
        cvec = _as(correction, np.float32).reshape(-1)
        f = self._cycle_fn("backprop")
        head = [C.c_void_p] if self._cycle_handle else []
        tail = [] if self._cycle_handle else [_f64p]
        f.argtypes = head + [C.c_int64, _f32p, _f32p, C.c_int32, C.c_int32, _f32p] + tail
        f.restype = C.c_int
        args = ([self._h] if self._cycle_handle else []) + [n, _ptr(poses, _f32p), _ptr(cov, _f32p), int(lo), int(hi), _ptr(cvec, _f32p)]
        if not self._cycle_handle:
            args.append(None)
        self._cycle_check(f(*args), "backprop")
        return poses.reshape(-1, 3), cov.reshape(-1, 9)

    def normalize_angles(self, poses_xyt):
        poses = np.array(poses_xyt, np.float32).reshape(-1).copy()
        f = self._cycle_fn("normalize")
        f.argtypes = [C.c_int64, _f32p]
        f.restype = None if not self._cycle_handle else C.c_int
        f(poses.size // 3, _ptr(poses, _f32p))
        return poses.reshape(-1, 3)

    def run_correction(self, ctype, selected_points_xy8, poses_xyt, cov9, constraint_groups, mode=abi.MODE_ODOM_HC,
                       select_threshold=0.05, stf_window=3, find_corr_kwargs=None, reproject=True):
        """One whole HitLSLAM::Run (HitLSLAM.cpp:380-480) on this backend.  constraint_groups is the session's list of
        row arrays (human_constraints_); returns dict(poses, cov9, constraint_groups, summary, status, ...).
        reproject=True (the default: the reference's Run always ends with transformPointCloudsToWorldFrame, HitLSLAM.cpp:476-479)
        adds out["world"] = world_clouds(poses, copy=False): VIEWS into the page-locked display buffers this object keeps (8 bytes
        per observation of pinned memory, no second host copy) -- valid until the next run_correction / world_clouds / set_map /
        close of this solver; a caller that keeps them longer copies them (np.array(...)).  reproject=False: poses and covariances
        only, for callers that do not display the map."""
        poses = np.array(poses_xyt, np.float32).reshape(-1, 3)
        cov = np.array(cov9, np.float32).reshape(-1, 9)
        pts = np.array(selected_points_xy8, np.float32).reshape(-1)
        out = dict(poses=poses, cov9=cov, constraint_groups=list(constraint_groups), summary=None, status="ok")
        seen = self.verify_points(poses, pts.reshape(-1, 2), select_threshold)
        if int(seen.sum()) != pts.size // 2:                    # verifyUserInput: every clicked point must be near a scan point
            out["status"] = "not verified"
            return out
        seg, inl, pas, its = self.fit_segments(poses, pts.reshape(2, 4))
        counts, _ = self.observation_sets(poses, seg.reshape(-1), 0.03, want_mask=False)
        od = self.order_user_input(counts, seg.reshape(-1))
        out.update(fitted=seg, order=od)
        lo, hi = od["bounds"]
        if not (lo >= 0 and hi >= 1):
            out["status"] = "selection overlap"
            return out
        poses, correction, rows = self.apply_explicit_correction(ctype, od["seg_xy8"], poses, od["corrected"], od["anchor"])
        out["constraint_groups"].append(rows)
        poses, cov = self.backprop(poses, cov, lo, hi, correction)
        poses = self.normalize_angles(poses)
        allrows = np.concatenate(out["constraint_groups"]) if out["constraint_groups"] else np.zeros(0, abi.HC_DTYPE)
        self.set_human_constraints(allrows)
        if mode != abi.MODE_ODOM_HC:     # the observation solve the reference keeps commented out (JointOptimization.cpp:1353-1373)
            self.find_correspondences(poses, stf_window, **(find_corr_kwargs or {}))
        poses, summ = self.solve(mode, poses)
        out.update(poses=poses, cov9=cov, summary=summ, correction=correction)
        if reproject:
            # transformPointCloudsToWorldFrame at the end of HitLSLAM::Run (HitLSLAM.cpp:476): the clouds land in page-locked
            # buffers this object keeps (hitl_host_alloc, allocated on first use): 3 ms instead of 10 at 20 M observations
            out["world"] = self.world_clouds(poses, copy=False)
        return out

    def world_clouds(self, poses_xyt, copy=True):
        """The display clouds (HitLSLAM::transformPointCloudsToWorldFrame): (world_x, world_y).  hitl_slam_amd.Solver lands
        them in page-locked staging buffers it keeps (sized by the resident map, dropped by set_map() / close()) and
        returns COPIES the caller owns.  copy=False returns views into the staging buffers instead (no 8-byte-per-observation
        copy): valid only until the next world_clouds() / set_map() / close() of this solver -- a view kept longer reads freed
        memory.  The base form is transform_scans."""
        if hasattr(self, "transform_scans_pinned"):
            n_obs = int(self.obs_offset[-1])
            if getattr(self, "_world_buf", None) is not None and getattr(self, "_world_buf_len", 0) < n_obs:
                self._drop_world_buf()                             # never write n_obs floats into a smaller block
            wx, wy, self._world_buf = self.transform_scans_pinned(poses_xyt, getattr(self, "_world_buf", None))
            self._world_buf_len = max(getattr(self, "_world_buf_len", 0), n_obs)
            return (np.array(wx), np.array(wy)) if copy else (wx, wy)
        return self.transform_scans(poses_xyt)

    def pose_covariances(self, mode, poses_xyt):
        """== ceres::Covariance + robot-frame rotation (vector_mapping.cpp:2772-2812): float64 [n,3,3]."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        assert pin.size == 3 * self.n_poses
        out = np.zeros(9 * self.n_poses, np.float64)
        f = getattr(self._lib, self._cov_name)
        f.argtypes = [C.c_void_p, C.c_int32, _f32p, _f64p]
        f.restype = C.c_int
        st = f(self._h, int(mode), _ptr(pin, _f32p), _ptr(out, _f64p))
        if st != abi.OK:
            raise HitlError(st, self._cov_name, self._detail())
        return out.reshape(-1, 3, 3)

    def replay_log(self, entries, poses_xyt, cov9, mode=abi.MODE_ODOM_HC, **kw):
        """== the replay loop of HitLSLAM_main.cpp:880-905 over hitl_slam_amd.io.load_input_log(path): one correction
        cycle per logged entry (HitLSLAM::replayLog), entries marked undone included, as in the reference.  Returns
        (poses, cov9, constraint_groups, [status per entry])."""
        poses = np.array(poses_xyt, np.float32).reshape(-1, 3)
        cov = np.array(cov9, np.float32).reshape(-1, 9)
        groups, statuses = [], []
        for e in entries:
            if e["type"] not in (2, 4, 5, 6) or len(e["points"]) != 4:
                statuses.append("unsupported")     # point / corner / unknown: "not supported yet" (EMinput.cpp:467-469)
                continue
            r = self.run_correction(e["type"], np.asarray(e["points"], np.float32).reshape(-1), poses, cov, groups, mode=mode, **kw)
            statuses.append(r["status"])
            if r["status"] == "ok":
                poses, cov, groups = r["poses"], r["cov9"], r["constraint_groups"]
        return poses, cov, groups, statuses

    # -- the path --------------------------------------------------------------------------
    def solve(self, mode, poses_xyt, allow_failure=False):
        """Returns (poses_out float32 [n,3], Summary)."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        assert pin.size == 3 * self.n_poses
        pout = np.empty_like(pin)
        summ = abi.Summary()
        allow = (abi.ERR_SOLVER_FAILURE,) if allow_failure else ()
        self._check(self._fn("solve")(self._h, int(mode), _ptr(pin, _f32p), _ptr(pout, _f32p), C.byref(summ)),
                    "solve", allow)
        return pout.reshape(-1, 3), summ

    def evaluate(self, mode, poses_xyt, want_gradient=True):
        pin = _as(poses_xyt, np.float32).reshape(-1)
        cost = C.c_double()
        g = np.zeros(3 * self.n_poses, np.float64) if want_gradient else None
        self._check(self._fn("evaluate")(self._h, int(mode), _ptr(pin, _f32p), C.byref(cost), _ptr(g, _f64p)),
                    "evaluate")
        return cost.value, g

    def linearize(self, mode, poses_ref_xyt, x, n_pairs=None):
        """Returns (cost, gradient[3n], pair_residuals[P,2], pair_jacobians[P,2,6])."""
        pref = _as(poses_ref_xyt, np.float32).reshape(-1)
        x = _as(x, np.float64).reshape(-1)
        if n_pairs is None:
            P, Cn = C.c_int64(), C.c_int64()
            self._check(self._fn("get_corr_counts")(self._h, C.byref(P), C.byref(Cn)), "get_corr_counts")
            n_pairs = P.value
        cost = C.c_double()
        g = np.zeros(3 * self.n_poses, np.float64)
        pr = np.zeros((max(n_pairs, 1), 2), np.float64)
        pj = np.zeros((max(n_pairs, 1), 2, 6), np.float64)
        self._check(self._fn("linearize")(self._h, int(mode), _ptr(pref, _f32p), _ptr(x, _f64p), C.byref(cost),
                                          _ptr(g, _f64p), _ptr(pr, _f64p), _ptr(pj, _f64p)), "linearize")
        return cost.value, g, pr[:n_pairs], pj[:n_pairs]


ALLGATHER_FN = C.CFUNCTYPE(C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64)


class CommOps(C.Structure):
    """== hitl_comm_ops: an all-gather on HOST buffers supplied by the caller."""
    _fields_ = [("allgather", ALLGATHER_FN), ("ctx", C.c_void_p)]


def torch_allgather_ops(group=None, device=None, counter=None):
    """hitl_comm_ops backed by torch.distributed (gloo on CPU tensors, or nccl == RCCL when
    `device` is a cuda device).  Returns (CommOps, keepalive).  counter: a two-element list that receives
    [exchanges, bytes sent] of this rank (tests: every rank of a solve must make the same number of exchanges)."""
    import torch
    import torch.distributed as dist
    world = dist.get_world_size(group)

    def _ag(ctx, send, recv, nbytes):
        try:
            src = torch.frombuffer((C.c_char * nbytes).from_address(send), dtype=torch.uint8).clone()
            if device is not None:
                src = src.to(device)
            out = torch.empty(world * nbytes, dtype=torch.uint8, device=src.device)
            dist.all_gather_into_tensor(out, src, group=group)
            C.memmove(recv, out.cpu().numpy().ctypes.data, world * nbytes)
            if counter is not None:
                counter[0] += 1
                counter[1] += int(nbytes)
            return 0
        except Exception as e:  # never let an exception cross the C boundary
            import sys
            sys.stderr.write("hitl allgather callback failed: %r\n" % (e,))
            return 1

    fn = ALLGATHER_FN(_ag)
    return CommOps(fn, None), fn


_lib_cache = {}


def load_library(path=None):
    """dlopen libhitlgpu.so.  Raises (never falls back) if it is missing."""
    path = path or LIB_PATH
    if path not in _lib_cache:
        if not os.path.exists(path):
            raise ImportError("%s not found: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                              "(hipcc --offload-arch=gfx950); there is no CPU fallback" % path)
        _lib_cache[path] = C.CDLL(path, mode=C.RTLD_GLOBAL)
    return _lib_cache[path]


def set_test_switch(name, value=1, lib_path=None):
    """hitl_test_set_switch (csrc/hitl_host.h): a TEST hook of the library, not part of its C ABI -- forces a code path small
    test maps do not reach by themselves (e.g. "stf_addr64", "cr_legacy", "kd_cap") for every Solver created afterwards in
    this process.  name=None clears every hook."""
    lib = load_library(lib_path)
    lib.hitl_test_set_switch.argtypes = [C.c_char_p, C.c_longlong]
    lib.hitl_test_set_switch.restype = C.c_int
    if lib.hitl_test_set_switch(None if name is None else name.encode(), int(value)) != 0:
        raise KeyError("hitl_test_set_switch: no test switch named %r" % (name,))


def apply_test_switches_from_env(var="HITL_TEST_SWITCHES"):
    """For test WORKER processes (tests/_submap_worker.py, tools/cr_size_sweep.py): 'name=value,name=value' in the named
    environment variable -> set_test_switch calls.  The library itself never reads that variable."""
    for item in filter(None, os.environ.get(var, "").split(",")):
        k, _, v = item.partition("=")
        set_test_switch(k.strip(), int(v or 1))


class Solver(SolverBase):
    """The MI355X backend (libhitlgpu.so)."""

    def transform_scans_device(self, poses_xyt):
        """hitl_solver_transform_scans_device: (device pointer of world x, of world y) as ints; the clouds stay in HBM."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        dx, dy = C.c_void_p(), C.c_void_p()
        f = self._lib.hitl_solver_transform_scans_device
        f.argtypes = [C.c_void_p, _f32p, C.POINTER(C.c_void_p), C.POINTER(C.c_void_p)]
        self._check(f(self._h, _ptr(pin, _f32p), C.byref(dx), C.byref(dy)), "solve")
        return dx.value, dy.value

    def transform_scans_pinned(self, poses_xyt, buffers=None):
        """hitl_solver_transform_scans into page-locked host memory from hitl_host_alloc (allocated once and returned for
        reuse): (world_x, world_y, buffers)."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        n_obs = int(self.obs_offset[-1])
        lib = self._lib
        lib.hitl_host_alloc.restype = C.c_void_p
        lib.hitl_host_alloc.argtypes = [C.c_size_t]
        if buffers is None:
            buffers = (lib.hitl_host_alloc(4 * max(n_obs, 1)), lib.hitl_host_alloc(4 * max(n_obs, 1)))
            if not buffers[0] or not buffers[1]:
                raise MemoryError("hitl_host_alloc failed")
        f = lib.hitl_solver_transform_scans
        f.argtypes = [C.c_void_p, _f32p, C.c_void_p, C.c_void_p]
        self._check(f(self._h, _ptr(pin, _f32p), buffers[0], buffers[1]), "solve")
        wx = np.ctypeslib.as_array(C.cast(buffers[0], C.POINTER(C.c_float)), shape=(max(n_obs, 1),))[:n_obs]
        wy = np.ctypeslib.as_array(C.cast(buffers[1], C.POINTER(C.c_float)), shape=(max(n_obs, 1),))[:n_obs]
        return wx, wy, buffers

    def host_free(self, buffers):
        self._lib.hitl_host_free.argtypes = [C.c_void_p]
        for b in buffers:
            self._lib.hitl_host_free(b)

    def get_info_mat(self, mode):
        """hitl_solver_get_info_mat: (rows, cols) int32 -- the reference's info_mat_ adjacency as a sorted upper-triangle COO list."""
        f = self._lib.hitl_solver_get_info_mat
        f.argtypes = [C.c_void_p, C.c_int32, C.POINTER(C.c_int64), C.c_void_p, C.c_void_p]
        ne = C.c_int64(0)
        self._check(f(self._h, int(mode), C.byref(ne), None, None), "solve")
        r, c = np.zeros(max(ne.value, 1), np.int32), np.zeros(max(ne.value, 1), np.int32)
        self._check(f(self._h, int(mode), C.byref(ne), r.ctypes.data_as(C.c_void_p), c.ctypes.data_as(C.c_void_p)), "solve")
        return r[:ne.value], c[:ne.value]

    def get_kd_trees(self):
        """(pts float32 [n_obs, 4], meta uint32 [n_obs]) of hitl_solver_get_kd_trees: every scan's KD-tree in the in-order layout."""
        n_obs = int(self.obs_offset[-1] - self.obs_offset[0])
        pts = np.zeros((max(n_obs, 1), 4), np.float32)
        meta = np.zeros(max(n_obs, 1), np.uint32)
        f = self._lib.hitl_solver_get_kd_trees
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        self._check(f(self._h, pts.ctypes.data_as(C.c_void_p), meta.ctypes.data_as(C.c_void_p)), "get_kd_trees")
        return pts[:n_obs], meta[:n_obs]

    _names = dict(create="hitl_solver_create", destroy="hitl_solver_destroy", set_map="hitl_solver_set_map",
                  set_hc="hitl_solver_set_human_constraints", set_corr="hitl_solver_set_correspondences",
                  get_corr_counts="hitl_solver_get_correspondence_counts",
                  get_corr="hitl_solver_get_correspondences", solve="hitl_solver_solve",
                  evaluate="hitl_solver_evaluate", linearize="hitl_solver_linearize",
                  find_corr="hitl_solver_find_correspondences")
    _scan_names = dict(transform="hitl_solver_transform_scans", verify="hitl_solver_verify_points",
                       obs_sets="hitl_solver_observation_sets")
    _cycle_names = dict(fit="hitl_solver_fit_segments", order="hitl_order_user_input",
                        explicit="hitl_solver_apply_explicit_correction", backprop="hitl_solver_backprop",
                        normalize="hitl_normalize_angles")
    _cycle_handle = True
    _cov_name = "hitl_solver_pose_covariances"

    def __init__(self, options=None, lib_path=None):
        lib = load_library(lib_path)
        lib.hitl_last_error.restype = C.c_char_p
        lib.hitl_status_string.restype = C.c_char_p
        lib.hitl_status_string.argtypes = [C.c_int]
        super().__init__(lib, options)

    def _bind(self):
        super()._bind()
        f = self._fn("find_corr")
        f.argtypes = [C.c_void_p, _f32p, C.c_int64]
        f.restype = C.c_int

    def _detail(self):
        try:
            return "%s" % self._lib.hitl_last_error().decode()
        except Exception:
            return ""

    # -- pose-range submaps (world_size > 1) -----------------------------------------------
    def set_partition(self, global_begin, n_global, halo_poses):
        """Submap rank: owns GLOBAL poses from global_begin; set_map then takes the scans of
        [global_begin - halo, end + halo) (no lower halo on rank 0, no upper halo on the last rank)."""
        f = self._lib.hitl_solver_set_partition
        f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int32]
        f.restype = C.c_int
        st = f(self._h, int(global_begin), int(n_global), int(halo_poses))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_set_partition", self._detail())
        self._n_global = int(n_global)

    def set_partition_range(self, global_begin, global_end, n_global, halo_poses):
        """Submap rank owning GLOBAL poses [global_begin, global_end); halos clipped at the map's ends; halo_poses < 0 =
        replicated scans (set_map then takes the WHOLE map, pose pairs may reach any distance across the cuts)."""
        f = self._lib.hitl_solver_set_partition_range
        f.argtypes = [C.c_void_p, C.c_int64, C.c_int64, C.c_int64, C.c_int32]
        f.restype = C.c_int
        st = f(self._h, int(global_begin), int(global_end), int(n_global), int(halo_poses))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_set_partition_range", self._detail())
        self._n_global = int(n_global)

    def comm_selftest(self):
        st = self._lib.hitl_solver_comm_selftest(self._h)
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_comm_selftest", self._detail())

    def evaluate_jacobian(self, poses_xyt):
        """Problem::Evaluate of the scan-to-scan problem (JointOptimization.cpp:1247-1252): (cost, residuals[2P],
        scipy-style CRS (indptr[2P+1], indices[12P], data[12P]) over 3*n_poses columns)."""
        pin = _as(poses_xyt, np.float32).reshape(-1)
        P, _ = self.correspondence_counts()
        res = np.zeros(2 * P, np.float64)
        rows = np.zeros(2 * P + 1, np.int32)
        cols = np.zeros(12 * P, np.int32)
        vals = np.zeros(12 * P, np.float64)
        cost = C.c_double()
        f = self._lib.hitl_solver_evaluate_jacobian
        f.argtypes = [C.c_void_p, _f32p, C.POINTER(C.c_double), _f64p, C.POINTER(C.c_int32), C.POINTER(C.c_int32), _f64p]
        f.restype = C.c_int
        st = f(self._h, _ptr(pin, _f32p), C.byref(cost), _ptr(res, _f64p), rows.ctypes.data_as(C.POINTER(C.c_int32)),
               cols.ctypes.data_as(C.POINTER(C.c_int32)), _ptr(vals, _f64p))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_evaluate_jacobian", self._detail())
        return cost.value, res, (rows, cols, vals)

    def set_timing_level(self, level):
        f = self._lib.hitl_solver_set_timing_level
        f.argtypes = [C.c_void_p, C.c_int32]
        f.restype = C.c_int
        st = f(self._h, int(level))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_set_timing_level", self._detail())

    def comm_ranks(self, want_kind=False):
        """Ranks the active transport holds (ncclCommCount for the built-in RCCL communicator)."""
        f = self._lib.hitl_solver_comm_ranks
        f.argtypes = [C.c_void_p, C.POINTER(C.c_int32), C.POINTER(C.c_int32)]
        f.restype = C.c_int
        n, k = C.c_int32(), C.c_int32()
        st = f(self._h, C.byref(n), C.byref(k))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_comm_ranks", self._detail())
        return (n.value, k.value) if want_kind else n.value

    def set_comm(self, ops, keepalive=None):
        f = self._lib.hitl_solver_set_comm
        f.argtypes = [C.c_void_p, C.POINTER(CommOps)]
        f.restype = C.c_int
        self._comm_keepalive = (ops, keepalive)
        st = f(self._h, C.byref(ops))
        if st != abi.OK:
            raise HitlError(st, "hitl_solver_set_comm", self._detail())

    def set_map(self, obs_offset, px, py, nx, ny, cov9=None):
        super().set_map(obs_offset, px, py, nx, ny, cov9)
        if getattr(self, "_n_global", None):
            self.n_poses = self._n_global       # solve()/find_correspondences() take GLOBAL pose arrays

    def find_correspondences(self, poses_xyt, window):
        pin = _as(poses_xyt, np.float32).reshape(-1)
        assert pin.size == 3 * self.n_poses, "poses_xyt must hold every pose of the (global) map: the C ABI takes a bare pointer"
        self._check(self._fn("find_corr")(self._h, _ptr(pin, _f32p), int(window)), "find_corr")


def device_copy_bandwidth(device=0, nbytes=1 << 30, repeats=10, lib_path=None):
    """hitl_device_copy_bandwidth: GB/s (read + write) of the library's float4 grid-stride copy over nbytes on `device`."""
    lib = load_library(lib_path)
    out = C.c_double()
    lib.hitl_device_copy_bandwidth.argtypes = [C.c_int32, C.c_int64, C.c_int32, C.POINTER(C.c_double)]
    st = lib.hitl_device_copy_bandwidth(int(device), int(nbytes), int(repeats), C.byref(out))
    if st != abi.OK:
        raise HitlError(st, "hitl_device_copy_bandwidth", "")
    return out.value


def device_memory(device=0, lib_path=None):
    """(free, total) bytes of a HIP device."""
    lib = load_library(lib_path)
    f, t = C.c_int64(), C.c_int64()
    lib.hitl_device_memory.argtypes = [C.c_int32, C.POINTER(C.c_int64), C.POINTER(C.c_int64)]
    st = lib.hitl_device_memory(int(device), C.byref(f), C.byref(t))
    if st != abi.OK:
        raise HitlError(st, "hitl_device_memory", "")
    return f.value, t.value


def comm_unique_id(lib_path=None):
    lib = load_library(lib_path)
    buf = C.create_string_buffer(128)
    st = lib.hitl_comm_unique_id(buf)
    if st != abi.OK:
        raise HitlError(st, "hitl_comm_unique_id")
    return buf.raw


def partition_range(obs_offset, rank, world_size, lib_path=None, align=1):
    """Pose range [begin, end) owned by `rank` (host-only; no GPU needed).  align > 1: cuts on multiples of `align` poses, which
    lets the ranks' linear solves take the separator form when `align` is a multiple of the pose band."""
    lib = load_library(lib_path)
    off = _as(obs_offset, np.int64)
    b, e = C.c_int64(), C.c_int64()
    lib.hitl_partition_range_aligned.argtypes = [C.c_int64, _i64p, C.c_int32, C.c_int32, C.c_int32, _i64p, _i64p]
    st = lib.hitl_partition_range_aligned(len(off) - 1, _ptr(off, _i64p), rank, world_size, int(align), C.byref(b), C.byref(e))
    if st != abi.OK:
        raise HitlError(st, "hitl_partition_range")
    return b.value, e.value
