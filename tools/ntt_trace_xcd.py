#!/usr/bin/env python3
"""Per-XCD view of the forward NTT's in-kernel trace (see tools/ntt_trace.py): does block b run on XCD b & 7, how do the
XCDs' finish times differ, and how does the per-generation workgroup lifetime evolve over the launch?"""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
bid = np.arange(len(a))
live = a[:, 0] > 0
a, bid = a[live], bid[live]
t = a[:, :5].astype(np.int64)
t -= t[:, 0].min()
us = 0.01
xcc = (a[:, 5] >> 32).astype(np.int64)
print("workgroups", len(a), "span %.1f us" % (t[:, 4].max() * us))
print("block & 7 == XCC_ID for %.2f %% of the workgroups" % (100.0 * np.mean((bid & 7) == xcc)))
for x in range(8):
    s = xcc == x
    if s.any():
        life = (t[s, 4] - t[s, 0]) * us
        print("  xcc %d: %5d wgs, first start %7.1f us, last end %8.1f us, mean lifetime %.1f us" % (
            x, s.sum(), t[s, 0].min() * us, t[s, 4].max() * us, life.mean()))
# lifetime and phase durations by start-time decile
order = np.argsort(t[:, 0])
for q in range(10):
    sel = order[q * len(order) // 10:(q + 1) * len(order) // 10]
    d = np.diff(t[sel], axis=1) * us
    print("  start decile %d: start %7.1f us  load %5.2f  rounds %5.2f  final %5.2f  lifetime %5.2f" % (
        q, t[sel, 0].mean() * us, d[:, 0].mean(), d[:, 1].mean(), d[:, 3].mean(), ((t[sel, 4] - t[sel, 0]) * us).mean()))
