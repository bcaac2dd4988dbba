#!/bin/bash
# GPU box: the un-profiled bench lines, the GPU test log and the full-size parity log of the current build -> gpurun_out/final/
mkdir -p gpurun_out/final
python -m pytest tests -m gpu -q 2>&1 | tail -6 > gpurun_out/final/gputests.txt
python -m pytest tests/test_gpu_fullsize.py tests/test_gpu_configs.py -m gpu -q -s 2>&1 | grep -v "^\.*$" | grep -v "^$" > gpurun_out/final/fullsize_parity.txt
python bench.py 2>/dev/null | tail -1 > gpurun_out/final/bench_c4.json
python bench.py --workload c2 --steps 20 --warmup 3 2>/dev/null | tail -1 > gpurun_out/final/bench_c2.json
cat gpurun_out/final/gputests.txt
python -c "
import json
for f in ('c4','c2'):
    j=json.load(open('gpurun_out/final/bench_%s.json'%f)); print(f, j['ms_per_step'], j['lm_iterations_per_resolve'], j['roofline']['frac'], j['roofline']['traffic'], (j.get('cpu_baseline') or {}).get('resolve_ms'), j.get('speedup_vs_cpu_port'))"
