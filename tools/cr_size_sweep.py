"""Development helper: LM solves over many map sizes, results as JSON lines (n, mode, iterations, termination, cost).
Run once as is and once with HITL_TEST_SWITCHES=cr_legacy=1 (the library's test hook: level-per-launch cyclic reduction) and diff: the two paths must agree on
iteration count and termination for every size (structural edge cases of the fused stages show up as FAILURE or as a
different iteration count).  usage: cr_size_sweep.py out.jsonl"""
import json, os, sys
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
import hitl_slam_amd as H
from conftest import Problem
from hitl_slam_amd.api import apply_test_switches_from_env
apply_test_switches_from_env()          # HITL_TEST_SWITCHES="cr_legacy=1": read by this script, not by the library

rng = np.random.default_rng(7)
sizes_a = sorted(set(list(range(1000, 1100, 3)) + list(range(16370, 16430, 5)) + list(range(20460, 20520, 5)) + list(range(32760, 33300, 37))
                     + [int(x) for x in rng.integers(20, 60000, 60)] + [17, 33, 39, 40, 41, 42, 640, 641, 655, 656, 657, 672, 673]))
sizes_c = sorted(set([int(x) for x in rng.integers(30, 30000, 30)] + [120, 123, 1966, 1967, 1968, 1969, 1970, 3070, 3073, 3100, 3121, 4610, 4611, 4660, 20000, 20001]))
if os.environ.get("HITL_SWEEP_WINDOW"):        # other band widths (B = 3 * window): sizes chosen per super-block count
    wv = int(os.environ["HITL_SWEEP_WINDOW"])
    sizes_a = []
    sizes_c = sorted(set([int(x) for x in rng.integers(30, 30000, 16)] + [wv * k + d for k in (40, 41, 656, 657, 1030, 1040, 1041, 1537, 6667) for d in (0, -1)]))
if os.environ.get("HITL_SWEEP_QUICK"):         # tests/test_gpu_fuzz.py: a few dozen sizes, seconds per pass
    sizes_a = sorted(set([int(x) for x in rng.integers(20, 40000, 24)] + [40, 41, 656, 657, 1030, 1041, 20480, 20481]))
    sizes_c = sorted(set([int(x) for x in rng.integers(30, 12000, 6)] + [123, 1968, 3100]))
out = open(sys.argv[1], "w")
for mode, sizes in ((0, sizes_a), (2, sizes_c)):
    for n in sizes:
        prob = Problem(n, 12 if not os.environ.get("HITL_SWEEP_WINDOW") else 48, ppf=200, groups=max(1, n // 1000), kind=0)
        s = H.Solver()
        prob.load(s)
        if mode == 2:
            s.find_correspondences(prob.init, int(os.environ.get("HITL_SWEEP_WINDOW", 3)))
        try:
            _, sm = s.solve(mode, prob.init, allow_failure=True)
            rec = dict(n=n, mode=mode, it=sm.iterations, term=sm.termination, cost=sm.final_cost)
        except Exception as ex:      # noqa
            rec = dict(n=n, mode=mode, err=str(ex)[:80])
        out.write(json.dumps(rec) + "\n"); out.flush()
        s.close()
print("done", len(sizes_a), len(sizes_c))
