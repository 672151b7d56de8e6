#!/usr/bin/env python3
"""Static instruction mix per kernel from a gfx950 assembly file (hipcc --cuda-device-only -S): tools/isa_counts.py file.s [substring]"""
import re
import subprocess
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for m in re.finditer(r"^(_ZN7sealhip\S+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    name, body = m.group(1), m.group(2)
    d = subprocess.run(["c++filt", name], capture_output=True, text=True).stdout.strip()
    mm = re.search(r"(\w+_kernel(<[^(]*>)?)\(", d)
    short = mm.group(1) if mm else d[:70]
    if pat not in short:
        continue
    lines = [l.strip() for l in body.split("\n") if l.strip() and not l.strip().startswith((";", "."))]
    n = lambda *p: sum(1 for l in lines if l.startswith(p))
    print("%-46s total %5d valu %5d mult %5d vmem_ld %4d vmem_st %4d ds %4d smem %4d waitcnt %4d nop %4d scratch %3d" % (
        short, len(lines), n("v_"), n("v_mad_u64", "v_mul_lo", "v_mul_hi", "v_mad_u32"), n("global_load", "buffer_load", "flat_load"),
        n("global_store", "buffer_store", "flat_store"), n("ds_"), n("s_load", "s_buffer_load"), n("s_waitcnt"), n("s_nop"),
        n("scratch_")))
