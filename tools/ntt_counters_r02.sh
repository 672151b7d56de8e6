#!/bin/bash
# SQ counters of the single-pass NTT kernels at N = 2^15, integer (55-bit primes, tools/ntt_only.py) and FP64 (50-bit primes,
# tools/ntt_ab.py) instances, one counter per rocprofv3 pass (no trace domains with --pmc). Runs on the GPU box via gpurun:
#   tools/ntt_counters_r02.sh > gpurun_out/ntt_counters_r02.txt
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES SQ_BUSY_CYCLES; do
  for which in int fp64; do
    rm -rf gpurun_out/pm && mkdir -p gpurun_out/pm
    if [ $which = int ]; then
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pm -o p -- python3 tools/ntt_only.py --logn 15 --polys 256 --reps 1 > /dev/null 2>&1
      rows=$((256*7))
    else
      rocprofv3 --pmc $c --output-format csv -d gpurun_out/pm -o p -- python3 tools/ntt_ab.py 15 168 > /dev/null 2>&1
      rows=$((168*12))
    fi
    python3 - "$which" "$c" "$rows" <<PY
import csv, sys
rows = [r for r in csv.DictReader(open("gpurun_out/pm/p_counter_collection.csv"))]
for kern in ("ntt_fwd_half", "ntt_inv_half"):
    v = [float(r["Counter_Value"]) for r in rows if kern in r["Kernel_Name"]]
    if v:
        print("%-5s %-13s %-22s %.5g per launch = %.5g per row (%d launches)" % (sys.argv[1], kern, sys.argv[2], sum(v) / len(v), sum(v) / len(v) / int(sys.argv[3]), len(v)))
PY
  done
done
rm -rf gpurun_out/pm
