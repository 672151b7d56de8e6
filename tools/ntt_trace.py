#!/usr/bin/env python3
"""Phase timeline of the forward single-pass NTT from in-kernel wall-clock stamps (measurement-only build:
make -C gemini-seal_amd exp; SEALHIP_LIBRARY=.../libsealhip_exp.so SEALHIP_NTT_TRACE=file python tools/ntt_only.py ...).
Stamps per workgroup: 0 start, 1 loads+top layer done, 2 rounds 1-3 + exchanges done, 3 ticket seen, 4 final round
+ stores issued; word 5 = (XCC_ID << 32) | HW_ID. Clock = 100 MHz."""
import sys
import numpy as np

a = np.fromfile(sys.argv[1], dtype=np.uint64).reshape(-1, 8)
a = a[a[:, 0] > 0]
t = a[:, :5].astype(np.int64)
t0 = t[:, 0].min()
t -= t0
us = 0.01
print("workgroups %d, kernel span %.1f us" % (len(a), (t[:, 4].max()) * us))
names = ["load+top", "rounds1-3", "ticket wait", "final+store issue"]
d = np.diff(t, axis=1) * us
for i, nme in enumerate(names):
    print("  %-18s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % (nme, d[:, i].mean(), *np.percentile(d[:, i], [10, 50, 90])))
life = (t[:, 4] - t[:, 0]) * us
print("  %-18s mean %6.2f us  p10 %6.2f  p50 %6.2f  p90 %6.2f" % ("lifetime", life.mean(), *np.percentile(life, [10, 50, 90])))
if a[:, 7].any():
    cyc = (a[:, 7] - a[:, 6]).astype(np.float64)
    wall = (t[:, 4] - t[:, 0]).astype(np.float64) * 10e-9
    print("  shader clock seen by the workgroups: mean %.0f MHz (p10 %.0f, p90 %.0f)" % (
        (cyc / wall).mean() / 1e6, *np.percentile(cyc / wall / 1e6, [10, 90])))
hw = a[:, 5]
cu = ((hw >> 32) << 16) | ((hw & 0xFFFFFFFF) >> 8 & 0xFF)
ids, inv = np.unique(cu, return_inverse=True)
print("distinct (xcc, se/sh/cu) ids: %d" % len(ids))
# steady-state concurrency: how many WGs are in each phase at sample times
lo, hi = np.percentile(t[:, 0], 20), np.percentile(t[:, 4], 80)
samples = np.linspace(lo, hi, 400)
occ = np.zeros((len(samples), 4))
for i in range(4):
    for k, s in enumerate(samples):
        occ[k, i] = np.count_nonzero((t[:, i] <= s) & (s < t[:, i + 1]))
print("steady-state WGs in phase (mean over samples): " + ", ".join("%s %.0f" % (n, o) for n, o in zip(names, occ.mean(axis=0))),
      " total %.0f" % occ.sum(axis=1).mean())
# one CU's timeline
k = np.argmax(np.bincount(inv))
sel = np.where(inv == k)[0]
sel = sel[np.argsort(t[sel, 0])][:12]
print("timeline of one CU (us): start, load done, rounds done, ticket, end")
for j in sel:
    print("   ", " ".join("%8.2f" % (v * us) for v in t[j]))
