#!/bin/bash
# same-box A/B of the correction-cycle entry points (wall clock per C-ABI call, bench.py's correction_cycle leg): usage cycle_lib_ab.sh lib.so ... ("-" = in-tree)
cd "${GRAFT_REPO_ROOT:-/root/repo}"
for rep in 1 2; do
for lib in "$@"; do
  if [ "$lib" = "-" ]; then unset HITL_LIBHITLGPU; else export HITL_LIBHITLGPU=$(realpath $lib); fi
  python bench.py --no-cpu-baseline --steps 3 --warmup 1 2>/dev/null | python -c "
import sys, json
r = json.loads([l for l in sys.stdin if l.startswith('{')][-1]); c = r['correction_cycle']
print('%-36s interactive %.2f  pre-solve %.2f  verify %.2f fit %.2f obs %.2f explicit %.2f backprop %.2f  search %.2f  reproject %.2f' % ('$lib', r['interactive_correction_ms'], c['pre_solve_stages_ms'], c['verify_points_ms'], c['fit_segments_ms'], c['observation_sets_ms'], c['explicit_correction_ms'], c['backprop_ms'], r['correspondence_search_ms'], c['transform_scans_pinned_host_ms']))"
done
done
