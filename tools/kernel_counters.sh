#!/bin/bash
# SQ counters per kernel of the bench step (one counter per rocprofv3 pass, batch 256): tools/kernel_counters.sh "<counters>" [kernel substring]
cd /tmp && export TMPDIR=/tmp && cd $GRAFT_REPO_ROOT
for c in $1; do
  rm -rf gpurun_out/pmk && mkdir -p gpurun_out/pmk
  rocprofv3 --pmc $c --output-format csv -d gpurun_out/pmk -o p -- python3 bench.py --batch 256 --steps 1 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
  python3 - "$c" "${2:-}" <<PY
import csv, sys, collections
agg = collections.defaultdict(list)
for r in csv.DictReader(open("gpurun_out/pmk/p_counter_collection.csv")):
    n = r["Kernel_Name"]
    if "sealhip" not in n or sys.argv[2] not in n:
        continue
    agg[n.split("sealhip::(anonymous namespace)::")[1].split("(")[0]].append(float(r["Counter_Value"]))
for k, v in sorted(agg.items()):
    print("%-22s %-40s %.5g per launch (%d launches)" % (sys.argv[1], k, sum(v) / len(v), len(v)))
PY
done
