#!/bin/bash
# Runs on the GPU box (via gpurun): the round's rocprofv3 evidence for bench.py, written under gpurun_out/<tag>/ so that it is
# merged back; copy what is to be judged into profiles/<round>/ afterwards.
#   tools/collect_profiles.sh r03
set -o pipefail
tag=${1:-r04}
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
# 4. other configs: BASELINE lines 4 and 5 through bench.py itself (round 3), configs 1 and 2 through tools/bench_configs.py
python3 bench.py --config 4 > $out/bench_cfg4.jsonl 2> $out/bench_cfg4.err
python3 bench.py --config 5 > $out/bench_cfg5.jsonl 2> $out/bench_cfg5.err
python3 tools/bench_configs.py --only cfg1,cfg2 > $out/configs_1gpu.jsonl 2> $out/configs_1gpu.err
# 4b. the multi-rank path on the real backend with one rank (RCCL collectives incl. the gather to self)
python3 -m torch.distributed.run --nnodes=1 --nproc-per-node=1 --master-addr 127.0.0.1 --master-port 29579 bench.py --gpus 1 --force-dist --no-cpu-baseline --ntt-polys 0 > $out/bench_force_dist.jsonl 2> $out/bench_force_dist.err
python3 tools/ntt_only.py --polys 4096 > $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --polys 4096 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 14 --polys 1024 >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 14 --polys 1024 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 16 --polys 512 >> $out/ntt_only.txt 2>&1
python3 tools/ntt_only.py --logn 16 --polys 512 --inverse >> $out/ntt_only.txt 2>&1
python3 tools/step_profile.py 1024 > $out/step_profile_b1024.txt 2>&1
python3 tools/step_profile.py 1024 cfg3sq 2>&1 | grep "^cfg" >> $out/step_profile_b1024.txt
python3 tools/step_profile.py 1024 cfg4 2>&1 | grep "^cfg" > $out/step_profile_side_configs.txt
python3 tools/step_profile.py 1024 cfg4mul 2>&1 | grep "^cfg" >> $out/step_profile_side_configs.txt
python3 tools/step_profile.py 256 cfg5 2>&1 | grep "^cfg" >> $out/step_profile_side_configs.txt
python3 tools/step_profile.py 4096 cfg1 2>&1 | grep "^cfg" >> $out/step_profile_side_configs.txt
# 5. FP64 against integer NTT instances on 50-bit primes (same box, back to back)
for l in 14 15 16; do
  python3 tools/ntt_ab.py $l $((672*32768/(1<<l))) 2>&1 | grep logn | sed 's/^/fp64 /' >> $out/ntt_fp64_ab.txt
  SEALHIP_NTT_NO_FP64=1 python3 tools/ntt_ab.py $l $((672*32768/(1<<l))) 2>&1 | grep logn | sed 's/^/int  /' >> $out/ntt_fp64_ab.txt
done
# 6. kernel trace of the config-4 rotate step (what the per-tag profile does not cover shows up here)
cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $out/stats4 -o cfg4 -- python3 $root/tools/step_profile.py 1024 cfg4 > /dev/null 2>&1
find $out/stats4 -name "*kernel_stats.csv" -exec cp {} $out/cfg4_rotate_kernel_stats.csv \;
rm -rf $out/stats4
cd $root
rm -rf $out/stats
# 7. (round 4) STRICT mode through bench.py, and the quarter-row projection at N = 2^16 (needs lib/libsealhip_exp.so)
python3 bench.py --mode strict > $out/bench_strict.jsonl 2> $out/bench_strict.err
[ -f gemini-seal_amd/lib/libsealhip_exp.so ] && tools/n65536_projection.sh > $out/n65536_quarter_row_projection.txt 2>&1
ls -la $out
