#!/bin/bash
# CPU-side sanitizer pass (GPU sanitizers are not available on the pool): tools/sanitize_cpu.sh [outfile]
#  1. the oracle (the checker of every parity claim) built with gcc -fsanitize=address,undefined, under tests/test_oracle.py
#  2. the product library's HOST code (C ABI argument checks, wire parser, BLAKE2Xb seed expansion, host batching, bounds
#     recurrences; device code not instrumented) built by `make -C gemini-seal_amd san`, under tests/test_host.py
# Both runs halt on the first report; leak detection is off (the Python interpreter's own allocations drown it).
set -e -o pipefail
cd "$(dirname "$0")/.."
out=${1:-/dev/stdout}
mkdir -p oracle/_native
gcc -O1 -g -fPIC -std=gnu11 -ffp-contract=off -fsanitize=address,undefined -fno-sanitize-recover=undefined -fno-omit-frame-pointer \
    -shared -o oracle/_native/libsealref_san.so oracle/sealref.c -lm
make -s -C gemini-seal_amd -j6 san
crt=$(ls /opt/rocm/lib/llvm/lib/clang/*/lib/linux/libclang_rt.asan-x86_64.so | head -1)
{
    echo "== oracle, gcc ASan + UBSan (tests/test_oracle.py) =="
    SEALREF_LIBRARY=$PWD/oracle/_native/libsealref_san.so LD_PRELOAD="$(gcc -print-file-name=libasan.so) $(gcc -print-file-name=libubsan.so)" \
        ASAN_OPTIONS=detect_leaks=0 UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_oracle.py -x -q -m "not gpu" 2>&1 | tail -3
    echo "== product library host code, clang ASan + UBSan (tests/test_host.py) =="
    SEALHIP_LIBRARY=$PWD/gemini-seal_amd/lib/libsealhip_san.so LD_PRELOAD="$crt" ASAN_OPTIONS=detect_leaks=0:protect_shadow_gap=0 \
        UBSAN_OPTIONS=print_stacktrace=1:halt_on_error=1 python -m pytest tests/test_host.py -x -q -m "not gpu" 2>&1 | tail -3
} | tee "$out"
