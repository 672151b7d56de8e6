#!/bin/bash
# Round 4 A/B of the forward transform on one box, three interleaved rounds:
#   new      = level-2 quotient (zero-high pairs) + single-precision quotient in the canonical store
#   new_brt  = level-2 quotient, Barrett step in the canonical store (SEALHIP_NTT_CANON_BARRETT=1)
#   r03      = level-1 quotient + Barrett store: the round-3 kernel (tools/ab_build.sh apx1 -DSEALHIP_NTT_APX=1 -DSEALHIP_NTT_EXPERIMENT;
#              SEALHIP_NTT_CANON_BARRETT is an A/B knob: measurement builds only, so build the "new" side with tools/ab_build.sh too)
# usage: tools/ab_r04_fwd.sh [polys] [logn]
POLYS=${1:-8192}; LOGN=${2:-15}
R=$PWD/gemini-seal_amd/lib
for r in 1 2 3; do
  echo -n "new      "; SEALHIP_LIBRARY=$R/libsealhip.so python tools/ntt_only.py --logn $LOGN --polys $POLYS | cut -c1-90
  echo -n "new_brt  "; SEALHIP_NTT_CANON_BARRETT=1 SEALHIP_LIBRARY=$R/libsealhip.so python tools/ntt_only.py --logn $LOGN --polys $POLYS | cut -c1-90
  echo -n "r03      "; SEALHIP_NTT_CANON_BARRETT=1 SEALHIP_LIBRARY=$R/libsealhip_apx1.so python tools/ntt_only.py --logn $LOGN --polys $POLYS | cut -c1-90
done
