"""Batched in-place NTT throughput (canonical outputs) at one size: tools/ntt_ab.py logn [polys] -> rows/s, HBM fraction.
Primes: cfg2's (2^14), cfg4's (2^15) or cfg5's first 12 (2^16) -- all 50-bit, so the FP64 instances serve them
(SEALHIP_NTT_NO_FP64=1 selects the integer ones for an A/B)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd")); sys.path.insert(0, ROOT)
import torch, sealhip as S
from tools.bench_configs import mk, timed, P14, P15_12, P16
logn = int(sys.argv[1]); n = 1 << logn
pr = {14: P14, 15: P15_12, 16: P16[:12]}[logn]
P = int(sys.argv[2]) if len(sys.argv) > 2 else (1 << 24) // (n * len(pr)) * 4
dev = torch.device("cuda", 0)
ctx = S.Context(S.SCHEME_CKKS, logn, pr, 1, 0)
ctx.use_default_stream()
k = len(pr) - 1
x = mk(ctx, (P, len(pr), n), pr, dev)
fwd = timed(lambda: ctx.ntt_negacyclic_harvey(x, P, k, S.BASE_KEY), 10)
inv = timed(lambda: ctx.inverse_ntt_negacyclic_harvey(x, P, k, S.BASE_KEY), 10)
rows = P * len(pr)
print({"logn": logn, "rows": rows, "fwd_rows_per_s": round(rows / fwd), "inv_rows_per_s": round(rows / inv),
       "fwd_hbm_frac": round(rows * 16 * n / fwd / 8e12, 4), "inv_hbm_frac": round(rows * 16 * n / inv / 8e12, 4)})
