#!/bin/bash
# A/B helper: tools/ab_build.sh <name> [extra hipcc flags] -> gemini-seal_amd/lib/libsealhip_<name>.so built from the current
# sources with the extra flags (e.g. -DSEALHIP_FLOOR_VARIANT=1); select it at run time with SEALHIP_LIBRARY=<path>.
set -e
cd "$(dirname "$0")/../gemini-seal_amd"
name=$1; shift
mkdir -p build_$name lib
for f in hostmath.cpp engine.cpp pipeline.cpp api.cpp wire.cpp blake2xb.cpp hostbatch.cpp ntt.hip poly.hip rns.hip keyswitch.hip rlwe.hip ckks_encoder.hip; do
  ( /opt/rocm/bin/hipcc -x hip --offload-arch=gfx950 -O3 -std=c++17 -fPIC -Wall -Wno-unused-function -I../include "$@" -c csrc/$f -o build_$name/$f.o ) &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o lib/libsealhip_$name.so build_$name/*.o
echo built lib/libsealhip_$name.so
