"""Development helper (GPU box): Figure-8 stand-in, KD search on the GPU vs the oracle's method 1; prints where they differ."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))), "tests"))
import numpy as np
import figure8
import hitl_slam_amd as H
from hitl_slam_amd import io as hio
from oracle import hitl_oracle as O
d = figure8.make(160, 90)
path = "/tmp/fig8.stfs.covars"
hio.save_stfs_covars(path, "AMRL", 1455638506.25, d["init"], d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
g = hio.load_stfs_covars(path, fix_normals=True)
for window, kd in ((3, 0), (0, 0), (3, 1)):
    opt = H.default_options(); opt.kd_tree_search = kd
    gs, o = H.Solver(opt), O.OracleSolver()
    for s in (gs, o):
        s.set_map(g["obs_offset"], g["px"], g["py"], g["nx"], g["ny"], g["cov9"])
    gs.find_correspondences(g["poses"], window)
    o.find_correspondences(g["poses"], window, kd)
    a, b = gs.get_correspondences(), o.get_correspondences()
    da = {(int(i), int(j)): (a[3][a[2][p]:a[2][p + 1]], a[4][a[2][p]:a[2][p + 1]]) for p, (i, j) in enumerate(zip(a[0], a[1]))}
    db = {(int(i), int(j)): (b[3][b[2][p]:b[2][p + 1]], b[4][b[2][p]:b[2][p + 1]]) for p, (i, j) in enumerate(zip(b[0], b[1]))}
    print("window", window, "kd", kd, "gpu pairs", len(da), "oracle pairs", len(db))
    shown = 0
    for key in sorted(set(da) | set(db)):
        x, y = da.get(key), db.get(key)
        if x is None or y is None or not (np.array_equal(x[0], y[0]) and np.array_equal(x[1], y[1])):
            print(" pair", key, "gpu", None if x is None else len(x[0]), "oracle", None if y is None else len(y[0]))
            if x is not None and y is not None:
                sx, sy = dict(zip(x[0].tolist(), x[1].tolist())), dict(zip(y[0].tolist(), y[1].tolist()))
                for k in sorted(set(sx) | set(sy)):
                    if sx.get(k) != sy.get(k):
                        print("   k", k, "gpu ->", sx.get(k), "oracle ->", sy.get(k))
            elif x is not None:
                print("   gpu matches", list(zip(x[0].tolist(), x[1].tolist())))
            else:
                print("   oracle matches", list(zip(y[0].tolist(), y[1].tolist())))
            shown += 1
            if shown >= 6: break
    gs.close(); o.close()
