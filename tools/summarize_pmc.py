#!/usr/bin/env python3
"""Mean of each PMC counter per dispatch, per kernel, over the passes tools/profile_pmc.sh wrote."""
import csv
import glob
import os
import sys
from collections import defaultdict

out = sys.argv[1]
vals = defaultdict(lambda: defaultdict(list))
for f in glob.glob(os.path.join(out, "p*", "*", "*counter_collection.csv")):
    for row in csv.DictReader(open(f)):
        vals[row["Kernel_Name"].split("(")[0]][row["Counter_Name"]].append(float(row["Counter_Value"]))
keep = ["k_linearize_stf", "k_cr_level", "k_cr_top_lds", "k_assemble", "k_match", "k_cr_backward_tree", "k_linearize_stf_moments", "k_cr_chunk", "k_spmv", "k_kd_build"]
ctrs = sorted({c for k in vals for c in vals[k]})
print("# PMC counters, mean per dispatch (%s)\n" % out)
print("| kernel | dispatches | " + " | ".join(ctrs) + " |")
print("|---|---|" + "---|" * len(ctrs))
for k in sorted(vals):
    if not any(s in k for s in keep):
        continue
    n = max(len(v) for v in vals[k].values())
    print("| %s | %d | " % (k[:48], n) + " | ".join(("%.4g" % (sum(vals[k][c]) / len(vals[k][c])) if vals[k].get(c) else "-") for c in ctrs) + " |")
for k in vals:
    if "k_linearize_stf" in k and "moments" not in k:
        v = {c: sum(x) / len(x) for c, x in vals[k].items()}
        print("\n## k_linearize_stf, derived\n")
        if "SQ_WAVE_CYCLES" in v:
            for name in ("SQ_WAIT_ANY", "SQ_WAIT_INST_ANY", "SQ_ACTIVE_INST_ANY", "SQ_ACTIVE_INST_VALU"):
                if name in v:
                    print("* %s / SQ_WAVE_CYCLES = %.3f" % (name, v[name] / v["SQ_WAVE_CYCLES"]))
        if "TCC_HIT_sum" in v and "TCC_MISS_sum" in v:
            print("* L2 hit rate = %.3f" % (v["TCC_HIT_sum"] / (v["TCC_HIT_sum"] + v["TCC_MISS_sum"])))
        if "SQ_INSTS_VALU" in v and "SQ_WAVES" in v:
            print("* VALU instructions per wave = %.1f, VMEM reads per wave = %.1f, LDS per wave = %.1f" %
                  (v["SQ_INSTS_VALU"] / v["SQ_WAVES"], v.get("SQ_INSTS_VMEM_RD", 0) / v["SQ_WAVES"], v.get("SQ_INSTS_LDS", 0) / v["SQ_WAVES"]))
