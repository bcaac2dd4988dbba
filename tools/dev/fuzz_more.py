"""Development helper (GPU box): tests/test_gpu_fuzz.py::test_seeded_problem on N more seeded problems than the suite carries."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_fuzz as F
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 77)
N = int(sys.argv[1]) if len(sys.argv) > 1 else 60
bad = 0
for k in range(N):
    n = int(rng.integers(9, 400)); obs = int(rng.integers(24, 260)); ppf = int(rng.integers(8, n + 1))
    window = int(rng.integers(1, 5)); kind = int(rng.choice([0, 2, 4, 5, 6])); groups = int(rng.integers(0, 5))
    try:
        F.test_seeded_problem(n, obs, ppf, window, kind, groups, 20270000 + k)
    except AssertionError as e:
        bad += 1
        print("FAIL", (n, obs, ppf, window, kind, groups, 20270000 + k), str(e)[:200])
print("%d problems, %d failures" % (N, bad))
