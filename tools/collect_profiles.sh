#!/bin/bash
# Runs on the GPU box (via gpurun): the round's rocprofv3 evidence for bench.py, written under gpurun_out/<tag>/ so that it is
# merged back; copy what is to be judged into profiles/<round>/ afterwards.
#   tools/collect_profiles.sh r02
set -o pipefail
tag=${1:-r02}
root=${GRAFT_REPO_ROOT:-$PWD}
out=$root/gpurun_out/${tag}_profiles
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
# 1. kernel trace + stats of the default bench command (the program itself after --)
rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats -o bench -- python3 $root/bench.py --no-cpu-baseline > $out/bench_under_rocprof.jsonl 2> $out/bench_under_rocprof.err
find $out/stats -name "*kernel_stats.csv" -exec cp {} $out/bench_default_kernel_stats.csv \;
# 2. HBM traffic of the NTT kernels: separate --pmc passes (FETCH_SIZE, WRITE_SIZE), done by bench.py itself as child runs
cd $root && python3 bench.py --measure-traffic --steps 3 --warmup 1 > $out/bench_with_traffic.jsonl 2> $out/bench_with_traffic.err
cp $root/profiles/traffic.json $out/traffic.json 2>/dev/null
# 3. the plain default run (what the driver runs), with the CPU baseline
python3 bench.py > $out/bench_default.jsonl 2> $out/bench_default.err
# 4. other configs and the boundary measurements
python3 tools/bench_configs.py > $out/configs_1gpu.jsonl 2> $out/configs_1gpu.err
python3 tools/ntt_only.py --polys 4096 > $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --polys 4096 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 14 --polys 1024 >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 14 --polys 1024 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 16 --polys 512 >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 16 --polys 512 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/step_profile.py 1024 > $out/step_profile_b1024.txt 2>&1
rm -rf $out/stats
ls -la $out
