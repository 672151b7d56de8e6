#!/bin/bash
# A/B of library builds on the same box: tools/ab_ntt.sh "<ntt_only args>" lib1.so lib2.so ...   (three interleaved rounds)
ARGS=$1; shift
for r in 1 2 3; do
  for lib in "$@"; do
    echo -n "$(basename $lib)  "
    SEALHIP_LIBRARY=$PWD/$lib python tools/ntt_only.py $ARGS | cut -c1-75
  done
done
