#!/bin/bash
# quick perf probe on the GPU box: stf kernel avg ms + resolve ms for the default workload
python bench.py --steps ${1:-3} --warmup 1 --no-cpu-baseline ${2:-} 2>/dev/null | python -c "
import json,sys
j=json.loads(sys.stdin.read().strip().splitlines()[-1])
print('resolve_ms %.2f iters %d stf_ms %.4f frac %.3f linsolve_ms/iter %.3f match_ms %.1f' % (j['resolve_ms'], j['lm_iterations_per_resolve'], j['roofline']['avg_launch_ms'], j['roofline']['frac'], j['linear_solver_ms_per_resolve']/max(j['lm_iterations_per_resolve'],1), j['correspondence_search_ms']))"
