#!/bin/bash
# SQ counters of the forward single-pass NTT (one counter per pass; measurement-only build for the skip variants)
# usage: tools/ntt_counters.sh "<SEALHIP_NTT_SKIP values>" "<counters>"
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
export SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/libsealhip_exp.so
for skip in $1; do
  for c in $2; do
    rm -rf gpurun_out/pm && mkdir -p gpurun_out/pm
    SEALHIP_NTT_SKIP=$skip rocprofv3 --pmc $c --output-format csv -d gpurun_out/pm -o p -- python3 tools/ntt_only.py --logn 15 --polys 256 --reps 1 > /dev/null 2>&1
    python3 - "$skip" "$c" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open("gpurun_out/pm/p_counter_collection.csv")) if "fwd_half" in r["Kernel_Name"]]
v = [float(r["Counter_Value"]) for r in rows]
print("skip=%s %-24s %.4g per launch (%.4g per row)" % (sys.argv[1], sys.argv[2], sum(v) / len(v), sum(v) / len(v) / (256 * 7)))
PY
  done
done
