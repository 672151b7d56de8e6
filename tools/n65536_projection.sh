#!/bin/bash
# Round 4, VERDICT r03 item 2: what a QUARTER-ROW inverse at N = 2^16 would cost, measured on the kernels that exist.
# A quarter-row workgroup of a 2^16 ring is the N = 2^15 half-row shape (512 lanes x 32 coefficients, 14 on-chip layers, two
# workgroups per CU) -- the very kernel that transforms a half row of a 2^15 ring; per byte it costs what that kernel costs.
# The two layers it leaves undone are either a consumer's load step (in the pipelines) or, for a standalone transform, one
# more streaming pass over the row: the cost of today's ntt_inv_top pass (16 N bytes in and out, one product per word more).
# So:  projected standalone 2^16 = [2^15 half-row kernel alone, same bytes]  +  [2^16 streaming top pass]
# against the standalone 2^16 of rounds 1-3 (half-row kernel of 1024 lanes, one per CU, + the same streaming pass).
# The projection was then BUILT for standalone transforms (ntt_inv_half_kernel<15, .., QUARTER> + ntt_inv_top2_kernel); the
# script prints both forms and the 2^15 kernel the projection was made from.
# Needs the measurement-only build (make -C gemini-seal_amd exp): SEALHIP_NTT_WHOLE_ROW=0 makes the 2^15 inverse run its
# half-row kernel + top pass (two profile tags) instead of the whole-row form.
R=$PWD/gemini-seal_amd/lib
for kind in fp64 int; do
  if [ $kind = int ]; then export SEALHIP_NTT_NO_FP64=1; else unset SEALHIP_NTT_NO_FP64; fi
  echo "== $kind instances, 50-bit primes, same number of bytes per launch (7 x 2^29 B)"
  echo -n "2^16 half-row 1024 lanes + top pass (SEALHIP_NTT_QUARTER=0: the form before the quarter-row kernels): "; SEALHIP_NTT_QUARTER=0 SEALHIP_LIBRARY=$R/libsealhip_exp.so python tools/ntt_only.py --logn 16 --polys 1024 --inverse | cut -c1-400
  echo -n "2^16 quarter-row kernels + radix-4 pass (BUILT, the default since round 4): "; SEALHIP_LIBRARY=$R/libsealhip_exp.so python tools/ntt_only.py --logn 16 --polys 1024 --inverse | cut -c1-400
  echo -n "2^15 half-row kernel + top pass (WHOLE_ROW=0): "; SEALHIP_NTT_WHOLE_ROW=0 SEALHIP_LIBRARY=$R/libsealhip_exp.so python tools/ntt_only.py --logn 15 --bits50 --polys 2048 --inverse | cut -c1-400
  echo -n "2^16 forward today: "; SEALHIP_LIBRARY=$R/libsealhip_exp.so python tools/ntt_only.py --logn 16 --polys 1024 | cut -c1-200
  echo -n "2^15 forward (same bytes): "; SEALHIP_LIBRARY=$R/libsealhip_exp.so python tools/ntt_only.py --logn 15 --bits50 --polys 2048 | cut -c1-200
done
