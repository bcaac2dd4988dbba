"""hostcheck driver (CPU only): walks the C ABI of the sanitizer build of the library's HOST code (hitl_slam_amd/csrc/hostcheck/
libhitlgpu_hostcheck.so: g++ -fsanitize=address,undefined over every .cpp file, a stub device layer underneath) through the calls a
session makes -- create, set_map, constraints, search, pair tables, the three solve modes, evaluate, covariances, the correction-cycle
stages, info_mat, bad arguments -- so that every staging buffer, table builder and loop of the host side runs under the sanitizers.
No kernel runs under the stub: numerical results are meaningless and not looked at; what counts is that ASan / UBSan stay silent
(either aborts the process) and that no call crashes.  Run by tests/hostcheck_run.sh."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
assert "hostcheck" in os.environ.get("HITL_LIBHITLGPU", ""), "run through tests/hostcheck_run.sh"
import hitl_slam_amd as H          # noqa: E402  (loads the hostcheck library: HITL_LIBHITLGPU)
from hitl_slam_amd import abi      # noqa: E402
from hitl_slam_amd.api import HitlError, partition_range  # noqa: E402


def tolerant(fn, *a, **k):
    try:
        return fn(*a, **k)
    except HitlError as e:         # a status code is a fine outcome under the stub; a sanitizer report is not (it aborts)
        return e
    except AssertionError as e:
        return e


rng = np.random.default_rng(5)
calls = 0
for n, m, window in ((40, 24, 3), (7, 5, 1), (130, 16, 4), (64, 33, 0)):
    counts = rng.integers(max(1, m - 4), m + 1, n)
    off = np.concatenate([[0], np.cumsum(counts)]).astype(np.int64)
    O = int(off[-1])
    px, py = rng.normal(0, 3, O).astype(np.float32), rng.normal(0, 3, O).astype(np.float32)
    ang = rng.uniform(-3, 3, O)
    nx, ny = np.cos(ang).astype(np.float32), np.sin(ang).astype(np.float32)
    cov = np.tile(np.eye(3, dtype=np.float32).reshape(-1) * 1e-4, (n, 1))
    poses = np.stack([np.arange(n) * 0.3, rng.normal(0, 0.01, n), rng.normal(0, 0.01, n)], 1).astype(np.float32)
    for opts in ({}, dict(kd_tree_search=0), dict(max_band_poses=1), dict(stf_moments=1), dict(timing_level=2)):
        o = H.default_options()
        for k, v in opts.items():
            setattr(o, k, v)
        s = H.Solver(o)
        s.set_map(off, px, py, nx, ny, cov)
        rows = np.zeros(9, abi.HC_DTYPE)
        rows["type"] = [2, 4, 5, 6, 2, 4, 5, 6, 2]
        rows["constrained_pose_id"] = rng.integers(1, n, 9)
        rows["anchor_pose_id"] = rng.integers(0, n, 9)
        rows["delta_parallel"] = 0.1
        s.set_human_constraints(rows)
        tolerant(s.find_correspondences, poses, window)
        # a caller-provided pair list as well (host-side pair tables: build_pair_structures), including an off-band pair
        ps = np.array([0, 1, 2, min(5, n - 1)], np.int32)
        pt = np.array([1, 0, min(n - 1, 2 + 5), 0], np.int32)
        lens = np.array([12, 11, 13, 15])
        mo = np.concatenate([[0], np.cumsum(lens)]).astype(np.int64)
        i0 = np.concatenate([rng.integers(0, counts[a], l) for a, l in zip(ps, lens)]).astype(np.int32)
        i1 = np.concatenate([rng.integers(0, counts[b], l) for b, l in zip(pt, lens)]).astype(np.int32)
        tolerant(s.set_correspondences, ps, pt, mo, i0, i1)
        tolerant(s.get_correspondences)
        for mode in (H.MODE_ODOM_HC, H.MODE_STF_ONLY, H.MODE_JOINT):
            tolerant(s.solve, mode, poses, allow_failure=True)
            tolerant(s.evaluate, mode, poses)
            calls += 2
        tolerant(s.pose_covariances, H.MODE_ODOM_HC, poses)
        tolerant(s.get_info_mat, H.MODE_JOINT)
        tolerant(s.transform_scans, poses)
        tolerant(s.verify_points, poses, np.zeros((4, 2), np.float32), 0.05)
        tolerant(s.observation_sets, poses, np.array([0, 0, 1, 0, 0, 1, 1, 1], np.float32), 0.03, True)
        tolerant(s.fit_segments, poses, np.array([[0, 0, 1, 0], [0, 1, 1, 1]], np.float32))
        od = tolerant(s.order_user_input, np.tile(np.array([[3, 0]], np.int32), (n, 1)), np.zeros(8, np.float32))
        corrected, anchor = np.arange(n // 2, n // 2 + 3, dtype=np.int32), np.arange(0, 3, dtype=np.int32)
        for ctype in (2, 4, 5, 6):
            tolerant(s.apply_explicit_correction, ctype, np.array([0, 0, 1, 0, 0, 1, 1, 1], np.float32), poses, corrected, anchor)
        tolerant(s.backprop, poses, cov, 1, n - 2, np.array([0.1, 0.0, 0.01], np.float32))
        tolerant(s.normalize_angles, poses)
        # bad arguments: status codes, not crashes
        tolerant(s.solve, 7, poses)
        bad = rows.copy(); bad["constrained_pose_id"][0] = n + 5
        tolerant(s.set_human_constraints, bad)
        tolerant(s.set_correspondences, ps, pt, mo[:-1], i0, i1)
        tolerant(s.find_correspondences, poses[:-1].copy() if n > 1 else poses, window)
        s.close()
        calls += 20
# GPU-free entry points
for w in (1, 2, 3, 8):
    for r in range(w):
        partition_range(off, r, w)
        partition_range(off, r, w, align=12)
print("hostcheck driver: %d calls walked, sanitizers silent" % calls)
