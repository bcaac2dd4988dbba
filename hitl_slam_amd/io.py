"""ctypes mirror of the hitl_io_* entry points (include/hitl_gpu.h): the file formats either
side of the re-solve -- .stfs.covars reader (== loadPoseGraph, HitLSLAM_main.cpp:192-300) and
writer (== SaveStfsandCovars, vector_mapping_main.cpp:1855-1901), results writer
(== saveHitLResults, HitLSLAM_main.cpp:572-581), input-log reader/writer (== LoadLogFile / LogActivity,
HitLSLAM_main.cpp:676-822).  Host-only: works without a GPU."""
import ctypes as C

import numpy as np

from . import abi
from .api import HitlError, load_library, _as, _ptr, _f32p, _i64p


def _lib():
    L = load_library()
    if not getattr(L, "_io_bound", False):
        vp = C.c_void_p
        L.hitl_io_load_stfs_covars.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.hitl_posegraph_free.argtypes = [vp]
        L.hitl_posegraph_free.restype = None
        for name, rt in (("num_poses", C.c_int64), ("num_obs", C.c_int64), ("timestamp", C.c_double),
                         ("map_name", C.c_char_p), ("poses", _f32p), ("cov9", _f32p), ("obs_offset", _i64p),
                         ("px", _f32p), ("py", _f32p), ("nx", _f32p), ("ny", _f32p)):
            f = getattr(L, "hitl_posegraph_" + name)
            f.argtypes = [vp]
            f.restype = rt
        L.hitl_io_save_stfs_covars.argtypes = [C.c_char_p, C.c_char_p, C.c_double, C.c_int64, _f32p, _i64p, _f32p, _f32p,
                                               _f32p, _f32p, _f32p]
        L.hitl_io_save_results.argtypes = [C.c_char_p, C.c_int64, _f32p]
        L.hitl_io_load_input_log.argtypes = [C.c_char_p, C.c_int, C.POINTER(vp)]
        L.hitl_input_log_free.argtypes = [vp]
        L.hitl_input_log_free.restype = None
        L.hitl_input_log_num_entries.argtypes = [vp]
        L.hitl_input_log_num_entries.restype = C.c_int32
        L.hitl_input_log_entry.argtypes = [vp, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32), C.POINTER(C.c_int32),
                                           C.POINTER(_f32p)]
        L.hitl_io_save_input_log.argtypes = [C.c_char_p, C.c_int32, C.POINTER(C.c_uint32), C.POINTER(C.c_int32),
                                             C.POINTER(C.c_int32), _f32p]
        L._io_bound = True
    return L


def load_stfs_covars(path, fix_normals=False):
    """Returns dict(map_name, timestamp, poses[n,3], cov9[n,9], obs_offset[n+1], px, py, nx, ny) with
    scans in the robot frame, exactly as the reference loader produces them."""
    L = _lib()
    h = C.c_void_p()
    st = L.hitl_io_load_stfs_covars(str(path).encode(), int(bool(fix_normals)), C.byref(h))
    if st != abi.OK:
        raise HitlError(st, "hitl_io_load_stfs_covars", "cannot read %s" % path)
    try:
        n, o = L.hitl_posegraph_num_poses(h), L.hitl_posegraph_num_obs(h)

        def arr(fn, count, dt):
            return np.ctypeslib.as_array(fn(h), shape=(max(count, 1),))[:count].astype(dt, copy=True) if count else np.zeros(0, dt)
        out = dict(map_name=L.hitl_posegraph_map_name(h).decode(), timestamp=L.hitl_posegraph_timestamp(h),
                   poses=arr(L.hitl_posegraph_poses, 3 * n, np.float32).reshape(n, 3),
                   cov9=arr(L.hitl_posegraph_cov9, 9 * n, np.float32).reshape(n, 9),
                   obs_offset=arr(L.hitl_posegraph_obs_offset, n + 1, np.int64),
                   px=arr(L.hitl_posegraph_px, o, np.float32), py=arr(L.hitl_posegraph_py, o, np.float32),
                   nx=arr(L.hitl_posegraph_nx, o, np.float32), ny=arr(L.hitl_posegraph_ny, o, np.float32))
    finally:
        L.hitl_posegraph_free(h)
    return out


def save_stfs_covars(path, map_name, timestamp, poses, obs_offset, px, py, nx, ny, cov9=None):
    L = _lib()
    poses = _as(poses, np.float32).reshape(-1)
    off = _as(obs_offset, np.int64)
    px, py, nx, ny = (_as(a, np.float32) for a in (px, py, nx, ny))
    cov9 = _as(cov9, np.float32)
    st = L.hitl_io_save_stfs_covars(str(path).encode(), map_name.encode(), float(timestamp), len(off) - 1,
                                    _ptr(poses, _f32p), _ptr(off, _i64p), _ptr(px, _f32p), _ptr(py, _f32p),
                                    _ptr(nx, _f32p), _ptr(ny, _f32p), _ptr(cov9, _f32p))
    if st != abi.OK:
        raise HitlError(st, "hitl_io_save_stfs_covars", "cannot write %s" % path)


def save_results(path, poses):
    L = _lib()
    poses = _as(poses, np.float32).reshape(-1)
    st = L.hitl_io_save_results(str(path).encode(), poses.size // 3, _ptr(poses, _f32p))
    if st != abi.OK:
        raise HitlError(st, "hitl_io_save_results", "cannot write %s" % path)


def load_input_log(path, fix_parallel=False):
    """Returns a list of dict(type, undone, points float32 [k,2]) == the reference's input_log_ (vector<SingleInput>)."""
    L = _lib()
    h = C.c_void_p()
    st = L.hitl_io_load_input_log(str(path).encode(), int(bool(fix_parallel)), C.byref(h))
    if st != abi.OK:
        raise HitlError(st, "hitl_io_load_input_log", "cannot read %s" % path)
    out = []
    try:
        for i in range(L.hitl_input_log_num_entries(h)):
            t, u, k, p = C.c_uint32(), C.c_int32(), C.c_int32(), _f32p()
            L.hitl_input_log_entry(h, i, C.byref(t), C.byref(u), C.byref(k), C.byref(p))
            pts = np.ctypeslib.as_array(p, shape=(2 * k.value,)).astype(np.float32, copy=True).reshape(-1, 2) if k.value else np.zeros((0, 2), np.float32)
            out.append(dict(type=int(t.value), undone=int(u.value), points=pts))
    finally:
        L.hitl_input_log_free(h)
    return out


def save_input_log(path, entries):
    L = _lib()
    n = len(entries)
    types = (C.c_uint32 * max(n, 1))(*[int(e["type"]) for e in entries])
    undone = (C.c_int32 * max(n, 1))(*[int(e["undone"]) for e in entries])
    npts = (C.c_int32 * max(n, 1))(*[len(e["points"]) for e in entries])
    pts = np.concatenate([np.asarray(e["points"], np.float32).reshape(-1) for e in entries]) if n else np.zeros(0, np.float32)
    pts = np.ascontiguousarray(pts if pts.size else np.zeros(1, np.float32))
    st = L.hitl_io_save_input_log(str(path).encode(), n, types, undone, npts, _ptr(pts, _f32p))
    if st != abi.OK:
        raise HitlError(st, "hitl_io_save_input_log", "cannot write %s" % path)
