#!/bin/bash
# round 3 A/B, FP64 instances and N = 2^16: register transposition (default) against the LDS trip (swap2 build)
cd "$(dirname "$0")/.."
for r in 1 2; do
  for lib in libsealhip_swap2.so libsealhip.so; do
    for logn in 15 16; do
      echo -n "$lib fp  | "; SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/ntt_ab.py $logn | cut -c1-150
      echo -n "$lib int | "; SEALHIP_NTT_NO_FP64=1 SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/ntt_ab.py $logn | cut -c1-150
    done
    echo -n "$lib | "; SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/step_profile.py 512 cfg4 | cut -c1-300
    echo -n "$lib | "; SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/step_profile.py 128 cfg5 | cut -c1-300
  done
done
