#!/usr/bin/env python3
"""Summarise a tools/profile_bench.sh output directory: per-kernel time table from the
rocprofv3 kernel trace and per-launch HBM bytes of the dominant kernel from the PMC passes
(FETCH_SIZE doubled on gfx950 as /opt/skills/guides/MI355X_MICROARCH.md section HBM
prescribes for wide coalesced reads; FETCH_SIZE/WRITE_SIZE are in KiB)."""
import csv
import glob
import json
import os
import sys
from collections import defaultdict


def find(d, pat):
    r = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return r[0] if r else None


def main():
    out = sys.argv[1]
    res = {}
    kt = find(os.path.join(out, "trace"), "*kernel_trace.csv")
    print("# rocprofv3 summary (%s)\n" % out)
    if kt:
        agg = defaultdict(lambda: [0, 0.0, 1e30, 0.0])
        for row in csv.DictReader(open(kt)):
            name = row["Kernel_Name"].split("(")[0]
            dur = (int(row["End_Timestamp"]) - int(row["Start_Timestamp"])) / 1e3  # us
            a = agg[name]
            a[0] += 1; a[1] += dur; a[2] = min(a[2], dur); a[3] = max(a[3], dur)
        tot = sum(a[1] for a in agg.values())
        print("## kernel trace (--kernel-trace --stats)\n")
        print("| kernel | calls | total us | avg us | min us | max us | % |")
        print("|---|---|---|---|---|---|---|")
        for name, a in sorted(agg.items(), key=lambda kv: -kv[1][1]):
            print("| %s | %d | %.1f | %.2f | %.2f | %.2f | %.1f |" % (name[:70], a[0], a[1], a[1] / a[0], a[2], a[3], 100 * a[1] / tot))
            if "k_linearize_stf" in name and "moments" not in name and "k_stf_" not in name:   # the streaming kernel (any of its structures), not k_linearize_stf_moments
                res["stf_avg_us"] = a[1] / a[0]; res["stf_calls"] = a[0]
        print()
    for key, sub, ctr in (("fetch", "pmc_fetch", "FETCH_SIZE"), ("write", "pmc_write", "WRITE_SIZE")):
        cc = find(os.path.join(out, sub), "*counter_collection.csv")
        if not cc:
            continue
        vals = defaultdict(list)
        for row in csv.DictReader(open(cc)):
            if row.get("Counter_Name") == ctr:
                vals[row["Kernel_Name"].split("(")[0]].append(float(row["Counter_Value"]))
        print("## PMC %s (KiB per dispatch, raw)\n" % ctr)
        print("| kernel | dispatches | mean KiB | max KiB |")
        print("|---|---|---|---|")
        for name, v in sorted(vals.items(), key=lambda kv: -sum(kv[1]))[:12]:
            print("| %s | %d | %.1f | %.1f |" % (name[:70], len(v), sum(v) / len(v), max(v)))
            if "k_linearize_stf" in name and "moments" not in name and "k_stf_" not in name:   # the streaming kernel (any of its structures), not k_linearize_stf_moments
                res[key + "_kib"] = sum(v) / len(v)
        print()
    if "fetch_kib" in res:
        fetch_b = 2.0 * res["fetch_kib"] * 1024.0      # gfx950: FETCH_SIZE reads 1/2 of a wide coalesced stream
        write_b = res.get("write_kib", 0.0) * 1024.0
        res["hbm_bytes_per_launch"] = fetch_b + write_b
        print("## k_linearize_stf HBM traffic per launch\n")
        print("FETCH_SIZE x2 (gfx950 correction) = %.1f MB, WRITE_SIZE = %.1f MB, total = %.1f MB" % (fetch_b / 1e6, write_b / 1e6, (fetch_b + write_b) / 1e6))
    try:
        bj = json.loads(open(os.path.join(out, "bench_trace.json")).read().strip().splitlines()[-1])
        res["n_obs"] = bj["config"]["n_obs_per_gpu"]
        res["n_matches"] = bj["config"]["matches_per_gpu"]
        res["source"] = "tools/profile_bench.sh %s: rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of `bench.py %s`, FETCH_SIZE x2 (gfx950)" % (
            os.path.basename(out.rstrip("/")), " ".join(sys.argv[2:]) if len(sys.argv) > 2 else "--steps 3 --warmup 1 --no-cpu-baseline")
        res["bench_avg_launch_ms"] = bj["roofline"]["avg_launch_ms"]
        print("\nbench.py (same run, under the profiler): avg k_linearize_stf launch %.4f ms by HIP events; kernel trace avg %.4f ms"
              % (bj["roofline"]["avg_launch_ms"], res.get("stf_avg_us", 0) / 1e3))
    except Exception as e:
        print("\n(no bench json: %r)" % (e,))
    try:      # the source text of the profiled kernel: bench.py uses these counters only while hitl_kernels.hip still is that text
        import hashlib
        src = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "hitl_slam_amd", "csrc", "hitl_kernels.hip")
        res["kernel_source_sha"] = hashlib.sha256(open(src, "rb").read()).hexdigest()[:16]
    except Exception:
        pass
    json.dump(res, open(os.path.join(out, "traffic.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
