"""Absolute kernel times (HIP events) of one multiply+relinearize step at config 3 size: tools/step_profile.py [batch]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd")); sys.path.insert(0, ROOT)
import torch, bench, sealhip as S
from tools.bench_configs import mk
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
n, k, pr = 1 << 15, 7, bench.CFG3_PRIMES
ctx = S.Context(S.SCHEME_BFV, 15, pr, 1, 786433)
ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
ev = S.Evaluator(ctx)
x, y = mk(ctx, (B, 2, k, n), pr[:k], dev), mk(ctx, (B, 2, k, n), pr[:k], dev)
o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
rk = S.KSwitchKeys(ctx, mk(ctx, (k, 2, 8, n), pr, dev), n_digits=k, from_host=False)
def step():
    ev.multiply(x, 2, y, 2, k, B, o); ev.relinearize_inplace(o, 3, k, B, [rk])
step(); ctx.synchronize()
ctx.profile_enable(True)
for _ in range(3):
    step()
ctx.synchronize()
prof = ctx.profile_fetch()
print({t: round(v["ms"] / 3, 3) for t, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}, "total", round(sum(v["ms"] for v in prof.values()) / 3, 2))
