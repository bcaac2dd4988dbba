"""File formats either side of the path: the library's .stfs.covars reader
against a line-by-line Python restatement of loadPoseGraph (HitLSLAM_main.cpp:192-300), the
writer/reader round trip, the bug-compatible normal transform, and BASELINE config 1's shape
(Figure-8 file, no human constraints) through the oracle."""
import math
import os

import numpy as np
import pytest

import figure8
from hitl_slam_amd import abi
from hitl_slam_amd import io as hio

F = np.float32
_libm = __import__("ctypes").CDLL("libm.so.6")
_libm.cosf.restype = _libm.sinf.restype = __import__("ctypes").c_float
_libm.cosf.argtypes = _libm.sinf.argtypes = [__import__("ctypes").c_float]


def py_loader(path, fix_normals=False):
    """Restatement of loadPoseGraph in plain Python + float32 arithmetic."""
    lines = open(path).read().split("\n")
    name, ts = lines[0].split()[0], float(lines[1])
    poses, covs, off, P, N = [], [], [0], [], []
    cur = []

    def flush():
        t = poses[-1]
        c, s = F(_libm.cosf(float(F(-t[2])))), F(_libm.sinf(float(F(-t[2]))))     # Rotation2Df(-angle): fp32 libm
        for (wx, wy, wnx, wny) in cur:
            qx, qy = F(wx + (-t[0])), F(wy + (-t[1]))
            P.append((F(c * qx) - F(s * qy), F(s * qx) + F(c * qy)))
            mx, my = (wnx, wny) if fix_normals else (F(wnx + (-t[0])), F(wny + (-t[1])))
            N.append((F(c * mx) - F(s * my), F(s * mx) + F(c * my)))
        off.append(len(P))
        cur.clear()

    for ln in lines[2:]:
        parts = [x for x in ln.replace(",", " ").split()]
        if len(parts) != 16:
            break
        v = [F(float(x)) for x in parts]
        if not poses or v[0] != poses[-1][0] or v[1] != poses[-1][1] or v[2] != poses[-1][2]:
            if poses:
                flush()
            poses.append(v[:3])
            covs.append(v[7:16])
        cur.append((v[3], v[4], v[5], v[6]))
    if cur:
        flush()
    P, N = np.array(P, F).reshape(-1, 2), np.array(N, F).reshape(-1, 2)
    return dict(map_name=name, timestamp=ts, poses=np.array(poses, F), cov9=np.array(covs, F), obs_offset=np.array(off, np.int64),
                px=P[:, 0], py=P[:, 1], nx=N[:, 0], ny=N[:, 1])


@pytest.fixture(scope="module")
def fig8_file(tmp_path_factory):
    d = figure8.make(80, 48)
    path = str(tmp_path_factory.mktemp("fig8") / "figure8_standin.bag.stfs.covars")
    hio.save_stfs_covars(path, "AMRL", 1455638506.25, d["init"], d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
    return path, d


@pytest.mark.parametrize("fix", [False, True])
def test_reader_matches_python_restatement_bit_exactly(fig8_file, fix):
    path, _ = fig8_file
    got, exp = hio.load_stfs_covars(path, fix_normals=fix), py_loader(path, fix_normals=fix)
    assert got["map_name"] == exp["map_name"] == "AMRL" and got["timestamp"] == exp["timestamp"]
    for k in ("poses", "cov9", "obs_offset", "px", "py", "nx", "ny"):
        assert np.array_equal(got[k], exp[k]), k


def test_round_trip_and_bug_compatible_normals(fig8_file):
    path, d = fig8_file
    fixed = hio.load_stfs_covars(path, fix_normals=True)
    assert np.array_equal(fixed["obs_offset"], d["off"])
    assert np.abs(fixed["poses"] - d["init"]).max() <= 5.1e-5            # "%.4f"
    # theta is stored with "%.4f": 5e-5 rad x up to 10 m of range
    assert np.abs(fixed["px"] - d["px"]).max() <= 1e-3 and np.abs(fixed["py"] - d["py"]).max() <= 1e-3
    assert np.abs(np.hypot(fixed["nx"], fixed["ny"]) - 1).max() <= 2e-4   # rotated only: still unit
    assert np.abs(fixed["cov9"] - d["cov"]).max() <= 1e-6                 # "%f"
    compat = hio.load_stfs_covars(path, fix_normals=False)
    # the reference applies the POINT transform to normals => not unit away from 0
    assert np.abs(np.hypot(compat["nx"], compat["ny"]) - 1).max() > 0.5
    assert np.array_equal(compat["px"], fixed["px"])


def test_grouping_rule_and_empty_scan(tmp_path):
    path = str(tmp_path / "t.stfs.covars")
    poses = np.array([[0, 0, 0], [1, 0, 0.5], [1, 0, 0.5]], F)            # two identical consecutive poses merge
    off = np.array([0, 2, 2, 3], np.int64)                                # middle pose has an empty scan
    px, py = np.array([1, 2, 3], F), np.array([0.5, 0.5, 0.5], F)
    nx, ny = np.array([1, 1, 1], F), np.array([0, 0, 0], F)
    hio.save_stfs_covars(path, "m", 1.0, poses, off, px, py, nx, ny, None)
    txt = open(path).read().strip().split("\n")
    assert len(txt) == 2 + 4                                              # empty scan -> one fake point row
    assert txt[4].startswith("1.0000,0.0000,0.5000,1.0000,0.0000, ")      # the fake point sits at the pose
    g = hio.load_stfs_covars(path, fix_normals=True)
    assert len(g["poses"]) == 2 and g["obs_offset"].tolist() == [0, 2, 4]  # identical rows re-merged by the loader


def test_results_writer(tmp_path):
    path = str(tmp_path / "hitl_results.txt")
    poses = np.array([[1.5, -2.25, 0.125], [3, 4, -3.1]], F)
    hio.save_results(path, poses)
    assert open(path).read() == "1.500000 -2.250000 0.125000\n3.000000 4.000000 -3.100000\n"


def test_config1_shape_through_the_oracle(fig8_file):
    """BASELINE config 1: Figure8 file, no human constraints, CPU path: runs without a GPU and
    leaves the map where it is (the odometry measurements are re-derived from the current poses)."""
    from oracle import hitl_oracle as O
    path, _ = fig8_file
    g = hio.load_stfs_covars(path)
    o = O.OracleSolver()
    o.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
    out, summ = o.solve(abi.MODE_ODOM_HC, g["poses"])
    assert summ.termination == abi.CONVERGENCE and summ.num_hc_residuals == 0
    d = out - g["poses"]
    d[:, 2] = np.arctan2(np.sin(d[:, 2]), np.cos(d[:, 2]))    # CopyParams wraps theta (JointOptimization.cpp:388)
    assert np.abs(d).max() <= 2e-6


# ---- input log ("recording"): LoadLogFile / LogActivity, HitLSLAM_main.cpp:676-822 ---------------------------
def _py_load_log(path, fix_parallel=False):
    """Line-by-line restatement of LoadLogFile (stoi / stof on comma-separated fields)."""
    lines = open(path).read().split("\n")
    n = int(lines[0].strip())
    out, at = [], 1
    for _ in range(n):
        if at >= len(lines) or lines[at] == "":
            break
        f = lines[at].split(",")
        at += 1
        ctype, undone = int(f[0]), int(f[1])
        k, t = 0, 0
        if ctype == 1:
            k, t = 2, 1
        elif ctype == 3:
            k, t = 8, 3
        elif ctype in (2, 4, 5, 6):
            k = 4
            t = {2: 2, 4: 4, 5: 5}.get(ctype, 6 if fix_parallel else 0)      # '== 7' in the reference: a logged 6 is lost
        pts = []
        for _j in range(k):
            g = lines[at].split(",")
            at += 1
            pts.append([np.float32(float(g[0])), np.float32(float(g[1]))])
        out.append(dict(type=t, undone=undone, points=np.array(pts, np.float32).reshape(-1, 2)))
    return out


def test_input_log_round_trip_and_format(tmp_path):
    from hitl_slam_amd import io as HIO
    rng = np.random.default_rng(5)
    entries = [dict(type=4, undone=0, points=rng.normal(0, 30, (4, 2)).astype(np.float32)),
               dict(type=2, undone=1, points=rng.normal(0, 30, (4, 2)).astype(np.float32)),
               dict(type=6, undone=0, points=rng.normal(0, 30, (4, 2)).astype(np.float32)),
               dict(type=5, undone=0, points=rng.normal(0, 30, (4, 2)).astype(np.float32)),
               dict(type=1, undone=0, points=rng.normal(0, 30, (2, 2)).astype(np.float32)),
               dict(type=3, undone=0, points=rng.normal(0, 30, (8, 2)).astype(np.float32))]
    path = str(tmp_path / "session.log")
    HIO.save_input_log(path, entries)
    text = open(path).read().split("\n")
    assert text[0] == "6 "                                        # "%d \n"
    assert text[1] == "4, 0"                                      # "%d, %d\n"
    assert text[2] == "%.4f, %.4f" % tuple(entries[0]["points"][0])
    for fix in (False, True):
        got = HIO.load_input_log(path, fix_parallel=fix)
        ref = _py_load_log(path, fix_parallel=fix)
        assert len(got) == len(ref) == 6
        for a, b, e in zip(got, ref, entries):
            assert a["type"] == b["type"] and a["undone"] == b["undone"] == e["undone"]
            assert np.array_equal(a["points"].view(np.uint32), b["points"].view(np.uint32))
            assert np.abs(a["points"] - e["points"]).max() <= 5.1e-5          # "%.4f"
    assert [e["type"] for e in HIO.load_input_log(path)] == [4, 2, 0, 5, 1, 3]          # the parallel entry comes back typeless
    assert [e["type"] for e in HIO.load_input_log(path, True)] == [4, 2, 6, 5, 1, 3]


def test_input_log_truncated_and_garbage(tmp_path):
    from hitl_slam_amd import io as HIO
    p = tmp_path / "short.log"
    p.write_text("3 \n4, 0\n1.0, 2.0\n3.0, 4.0\n5.0, 6.0\n7.0, 8.0\n")     # announces 3, holds 1: the reader stops at EOF
    got = HIO.load_input_log(str(p))
    assert len(got) == 1 and got[0]["points"].shape == (4, 2)
    p.write_text("2 \n9, 0\n7, 1\n")                                        # unknown types carry no points
    got = HIO.load_input_log(str(p))
    assert [(e["type"], e["undone"], len(e["points"])) for e in got] == [(0, 0, 0), (0, 1, 0)]
    p.write_text("0 \n")
    assert HIO.load_input_log(str(p)) == []
    with pytest.raises(Exception):
        HIO.load_input_log(str(tmp_path / "missing.log"))
    p.write_text("x\n")
    with pytest.raises(Exception):
        HIO.load_input_log(str(p))


def test_posegraph_accessors_are_null_safe():
    """Every hitl_posegraph_* accessor answers NULL / 0 for a NULL handle instead of dereferencing it."""
    import ctypes as C
    from hitl_slam_amd.api import load_library
    L = load_library()
    for name in ("poses", "cov9", "obs_offset", "px", "py", "nx", "ny"):
        f = getattr(L, "hitl_posegraph_" + name)
        f.restype = C.c_void_p
        f.argtypes = [C.c_void_p]
        assert f(None) is None
    L.hitl_posegraph_num_poses.restype = C.c_int64
    L.hitl_posegraph_num_poses.argtypes = [C.c_void_p]
    assert L.hitl_posegraph_num_poses(None) == 0
