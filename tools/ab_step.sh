#!/bin/bash
# A/B of library builds inside the config-3 step (same box, interleaved): tools/ab_step.sh [batch] lib1.so lib2.so ...
cd "$(dirname "$0")/.."
B=$1; shift
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$lib | "
    SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/$lib python tools/step_profile.py $B cfg3 2>/dev/null | cut -c1-330
  done
done
