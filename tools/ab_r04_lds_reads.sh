#!/bin/bash
# Round 4 A/B: exchange reads as single ds_read_b64 (-DSEALHIP_NTT_LDS_READ_SINGLE=1 = libsealhip_lds1.so) against the compiler's
# paired ds_read2_b64 (libsealhip.so built without it); standalone transforms, then the config-3 step
R=$PWD/gemini-seal_amd/lib
A=${1:-libsealhip.so}; B=${2:-libsealhip_lds1.so}
for r in 1 2 3; do
  for l in $A $B; do echo -n "$l fwd "; SEALHIP_LIBRARY=$R/$l python tools/ntt_only.py --logn 15 --polys 8192 | cut -c1-90; done
done
for l in $A $B; do echo -n "$l inv "; SEALHIP_LIBRARY=$R/$l python tools/ntt_only.py --logn 15 --polys 8192 --inverse | cut -c1-90; done
for l in $A $B; do echo -n "$l fwd 2^16 "; SEALHIP_LIBRARY=$R/$l python tools/ntt_only.py --logn 16 --polys 1024 | cut -c1-90; done
for l in $A $B; do echo -n "$l fwd 2^14 "; SEALHIP_LIBRARY=$R/$l python tools/ntt_only.py --logn 14 --polys 4096 | cut -c1-90; done
tools/ab_step.sh 1024 $A $B | cut -c1-150
