"""Development helper (GPU box): GPU-built KD-trees vs the CPU harness build (reference rules) on the Figure-8 stand-in."""
import os, sys, ctypes as C, subprocess
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np
import figure8
import hitl_slam_amd as H
from hitl_slam_amd import io as hio
import test_kdwalk_cpu as T
os.makedirs(os.path.dirname(T.LIB), exist_ok=True)
subprocess.check_call(["g++", "-O2", "-std=c++17", "-fPIC", "-shared", "-ffp-contract=off", "-o", T.LIB, T.SRC])
lib = C.CDLL(T.LIB)
d = figure8.make(160, 90)
path = "/tmp/fig8.stfs.covars"
hio.save_stfs_covars(path, "AMRL", 1455638506.25, d["init"], d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
g = hio.load_stfs_covars(path, fix_normals=True)
s = H.Solver()
s.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
pts, meta = s.get_kd_trees()
off = g["obs_offset"]
nbad = 0
for i in range(len(off) - 1):
    a, b = off[i], off[i + 1]
    hp, hm = T.build_tree(lib, g["px"][a:b], g["py"][a:b], g["nx"][a:b], g["ny"][a:b])
    if not (np.array_equal(hm[:b - a], meta[a:b]) and np.array_equal(hp[:b - a].view(np.uint32), pts[a:b].view(np.uint32))):
        nbad += 1
        if nbad <= 3:
            w = np.nonzero(hm[:b - a] != meta[a:b])[0]
            print("scan", i, "differs at positions", w[:12], "harness", [(int(x) & 0xffff, int(x) >> 16) for x in hm[w[:6]]], "gpu", [(int(x) & 0xffff, int(x) >> 16) for x in meta[a:b][w[:6]]])
print("trees that differ:", nbad, "of", len(off) - 1)
