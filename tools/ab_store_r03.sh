#!/bin/bash
# round 3 A/B of the final-round store forms (same box, interleaved): tools/ab_store_r03.sh lib1 lib2 ...
# standalone integer forward NTT at N = 2^15 (config 3's 55-bit primes, 4096 polys x 7 rows), then the config-3 step profile
cd "$(dirname "$0")/.."
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib) | "
    SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/ntt_only.py --logn 15 --polys 4096 --reps 10 | cut -c1-110
  done
done
for r in 1 2; do
  for lib in "$@"; do
    echo -n "$(basename $lib) | "
    SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/step_profile.py 1024 cfg3 | cut -c1-330
  done
done
