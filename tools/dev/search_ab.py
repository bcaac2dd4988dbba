"""A/B of the all-pairs (source-major) tree search on bench.py --workload c3's map: with and without the occupancy grids
(test hook kd_grid), index sets compared, wall time of find_correspondences (median of `reps`).
Usage: python tools/dev/search_ab.py [n_poses points laps reps]"""
import os
import sys
import time

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(__file__), "..", ".."))
sys.path.insert(0, os.path.join(os.path.dirname(__file__), ".."))
import figure8  # noqa: E402
import hitl_slam_amd as H  # noqa: E402

n, m, laps, reps = ([int(x) for x in sys.argv[1:5]] + [1000, 360, 4, 5][len(sys.argv) - 1:])[:4]
d = figure8.make(n, m, seed=7, laps=laps)
out = {}
for grid, spec in ((1, 4), (1, 2), (1, 0), (0, 4), (0, 0)):
    H.set_test_switch("kd_grid", grid)
    H.set_test_switch("kd_spec", spec)
    s = H.Solver(H.default_options())
    s.set_map(d["off"], d["px"], d["py"], d["nx"], d["ny"], d["cov"])
    ts = []
    for _ in range(reps + 1):
        t0 = time.perf_counter()
        s.find_correspondences(d["init"], 0)
        ts.append(1e3 * (time.perf_counter() - t0))
    c = s.get_correspondences()
    s.close()
    print("kd_grid=%d kd_spec=%d: first call %.2f ms, then median %.2f ms (min %.2f); %d pairs, %d matches" % (grid, spec, ts[0], float(np.median(ts[1:])), min(ts[1:]), len(c[0]), len(c[3])))
    if out:
        assert all(np.array_equal(a, b) for a, b in zip(out[0], c)), "index sets differ"
    out[0] = c
print("index sets identical")
