#!/bin/bash
# same-box A/B of two builds of libhitlgpu.so: usage lib_ab.sh <other .so> ; kernel-trace averages of the cyclic-reduction and linearisation kernels
cd "${GRAFT_REPO_ROOT:-/root/repo}"
export TMPDIR=/tmp
OTHER=$(realpath $1)
for rep in 1 2; do
for tag in head other; do
  OUT=gpurun_out/libab_$tag
  rm -rf $OUT; mkdir -p $OUT
  if [ "$tag" = "other" ]; then export HITL_LIBHITLGPU=$OTHER; else unset HITL_LIBHITLGPU; fi
  timeout 300 rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/t -- python3 tools/dev/solve_ac.py > $OUT/out.txt 2> $OUT/err.txt < /dev/null
  echo "== $tag"; cat $OUT/out.txt
  python3 - <<PY
import csv, glob
for f in glob.glob("$OUT/t/*/*kernel_stats.csv"):
    for r in csv.DictReader(open(f)):
        n = r["Name"]
        if "k_cr_chunk" in n or "k_cr_seq" in n or "k_host_flag" in n or "k_linearize_stf" in n or "k_assemble" in n or "k_spmv" in n:
            print("   %-44s calls %5s avg %8.2f us" % (n.split("(")[0].replace("void hitl::","").replace("hitl::","")[:44], r["Calls"], float(r["AverageNs"]) / 1e3))
PY
done
done
