#!/usr/bin/env python3
"""Print VGPR/SGPR/spill/scratch/LDS per kernel from a gfx950 assembly file (hipcc --cuda-device-only -S)."""
import re
import sys

txt = open(sys.argv[1]).read()
pat = sys.argv[2] if len(sys.argv) > 2 else ""
for blk in re.split(r"\n  - \.agpr_count:", txt)[1:]:
    def g(k):
        m = re.search(r"\.%s:\s*(\S+)" % k, blk)
        return m.group(1) if m else "?"
    name = g("name")
    if pat in name:
        print(f"{name[:90]:90s} vgpr {g('vgpr_count'):>4s} spill {g('vgpr_spill_count'):>3s} sgpr {g('sgpr_count'):>3s} scratch {g('private_segment_fixed_size'):>4s} lds {g('group_segment_fixed_size')}")
