#!/bin/bash
# Times the forward single-pass NTT with parts of the kernel removed (measurement-only build: make -C gemini-seal_amd exp).
# 0x100 = no butterflies in the rounds, 0x200 = no LDS exchanges, 0x400 = no global loads / top layer,
# 0x800 = no global stores. Results are wrong by construction; only time matters.
export SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/libsealhip_exp.so
L=${1:-15}
for skip in ${2:-0 0x100 0x200 0x300 0xC00 0xE00 0x400 0x800 0xD00}; do
  echo -n "skip=$skip  "
  SEALHIP_NTT_SKIP=$skip python tools/ntt_only.py --logn $L | cut -c1-110
done
