#!/usr/bin/env python3
"""NTT-only driver for profiling: forward (and optionally inverse) NTT over a batch at N=2^logn."""
import argparse, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd"))
import numpy as np
import sealhip as S

ap = argparse.ArgumentParser()
ap.add_argument("--logn", type=int, default=15)
ap.add_argument("--polys", type=int, default=1024)
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--inverse", action="store_true")
ap.add_argument("--bits50", action="store_true", help="logn 15 with config 4's 50-bit primes (FP64 instances) instead of config 3's")
a = ap.parse_args()
PR = {15: [36028797010444289, 36028797012606977, 36028797013000193, 36028797013327873, 36028797014376449,
           36028797014573057, 36028797014704129, 36028797017456641],
      14: [1125899903107073, 1125899903500289, 1125899903795201, 1125899903827969, 1125899903991809, 1125899904679937],
      16: [1125899864506369, 1125899865948161, 1125899870011393, 1125899870404609, 1125899877875713]}[a.logn]
if a.bits50 and a.logn == 15:
    PR = [1125899885412353, 1125899885740033, 1125899886395393, 1125899887312897, 1125899896160257, 1125899899174913,
          1125899901665281, 1125899902124033]
ctx = S.Context(S.SCHEME_CKKS, a.logn, PR, 1, 0)
k, n = len(PR) - 1, 1 << a.logn
rng = np.random.default_rng(0)
x = np.stack([rng.integers(0, p, size=(a.polys, n), dtype=np.uint64) for p in PR[:k]], axis=1)
d = ctx.upload(x)
fn = ctx.inverse_ntt_negacyclic_harvey if a.inverse else ctx.ntt_negacyclic_harvey
fn(d, a.polys, k); ctx.synchronize()
ctx.profile_enable(True)
t0 = time.perf_counter()
for _ in range(a.reps):
    fn(d, a.polys, k)
ctx.synchronize()
dt = time.perf_counter() - t0
prof = ctx.profile_fetch()
rows = a.polys * k * a.reps
ms = sum(v["ms"] for v in prof.values())
print("logn %d %s: %.3f M NTT/s wall, %.3f M NTT/s kernel, %.1f%% of HBM roofline (16N B/row)  %s" % (
    a.logn, "inv" if a.inverse else "fwd", rows / dt / 1e6, rows / (ms / 1e3) / 1e6,
    rows * 16 * n / (ms / 1e3) / 8e12 * 100, prof))
