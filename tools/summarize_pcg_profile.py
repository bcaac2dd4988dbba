#!/usr/bin/env python3
"""Per-launch HBM-side bytes of k_pcg_fused from a tools/profile_bench.sh output directory of `bench.py --workload c3`
(FETCH_SIZE x2 by the gfx950 rule of /opt/skills/guides/MI355X_MICROARCH.md, WRITE_SIZE as read; both in KiB per dispatch) ->
<dir>/traffic_c3.json, which bench.py uses for `roofline.traffic` of that workload while hitl_pcg.hip is the profiled text.
usage: summarize_pcg_profile.py gpurun_out/prof_r06c3"""
import csv, glob, json, os, sys
from collections import defaultdict
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

out = sys.argv[1]
res = {}
for key, sub, ctr in (("fetch", "pmc_fetch", "FETCH_SIZE"), ("write", "pmc_write", "WRITE_SIZE")):
    v = []
    for f in glob.glob(os.path.join(out, sub, "**", "*counter_collection.csv"), recursive=True):
        for row in csv.DictReader(open(f)):
            if row.get("Counter_Name") == ctr and "k_pcg_fused" in row["Kernel_Name"]: v.append(float(row["Counter_Value"]))
    res[key + "_kib"] = sum(v) / max(len(v), 1); res[key + "_dispatches"] = len(v)
dur = []
for f in glob.glob(os.path.join(out, "trace", "**", "*kernel_trace.csv"), recursive=True):
    for row in csv.DictReader(open(f)):
        if "k_pcg_fused" in row["Kernel_Name"]: dur.append((int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3)
res["pcg_avg_us"] = sum(dur) / max(len(dur), 1); res["pcg_calls"] = len(dur)
res["hbm_bytes_per_launch"] = 2.0 * res["fetch_kib"] * 1024.0 + res["write_kib"] * 1024.0
bj = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])
res["n_matches"] = bj["config"]["matches_per_gpu"]
res["cg_iterations_per_launch"] = bj["roofline"]["cg_iterations_per_launch"]
res["bench_us_per_cg_iteration"] = bj["roofline"]["us_per_cg_iteration"]
res["source"] = "tools/profile_bench.sh %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py --workload c3`, FETCH_SIZE x2 (gfx950)" % os.path.basename(out.rstrip("/"))
res["kernel_source_sha"] = bench.pcg_source_sha()
json.dump(res, open(os.path.join(out, "traffic_c3.json"), "w"), indent=1)
print("k_pcg_fused: %d launches, %.1f us each (kernel trace), FETCH_SIZE x2 %.2f MB + WRITE_SIZE %.2f MB = %.2f MB per launch of %.1f CG iterations (%.2f MB per CG iteration)" % (
    res["pcg_calls"], res["pcg_avg_us"], 2 * res["fetch_kib"] * 1024 / 1e6, res["write_kib"] * 1024 / 1e6, res["hbm_bytes_per_launch"] / 1e6,
    res["cg_iterations_per_launch"], res["hbm_bytes_per_launch"] / 1e6 / max(res["cg_iterations_per_launch"], 1e-9)))
