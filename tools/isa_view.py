#!/usr/bin/env python3
"""Compressed view of one kernel's ISA: memory instructions, waits, branches, labels; other instructions counted.
tools/isa_view.py file.s <mangled-name-substring> [max_lines]"""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2]
for m in re.finditer(r"^(_ZN7sealhip\S+):[^\n]*\n(.*?)^\.Lfunc_end\d+:", txt, re.S | re.M):
    if pat not in m.group(1):
        continue
    out, cnt = [], 0
    for l in m.group(2).split("\n"):
        t = l.strip()
        if not t or t.startswith(";"):
            continue
        if t.startswith(("s_load", "s_waitcnt", "global_", "scratch_", "ds_", "s_cbranch", "s_branch", "s_barrier")) or t.endswith(":"):
            if cnt:
                out.append("   ... %d other" % cnt)
                cnt = 0
            out.append(t[:120])
        else:
            cnt += 1
    if cnt:
        out.append("   ... %d other" % cnt)
    print("\n".join(out[: int(sys.argv[3]) if len(sys.argv) > 3 else 400]))
    break
