import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gemini-seal_amd")); sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, bench, sealhip as S
from tools.bench_configs import P15_12, mk, timed
dev = torch.device("cuda", 0)
logn, n = 15, 1 << 15
ctx = S.Context(S.SCHEME_CKKS, logn, P15_12, 1, 0)
ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
ev = S.Evaluator(ctx)
B, k = 512, 11
c = mk(ctx, (B, 2, k, n), P15_12[:k], dev)
key = mk(ctx, (k, 2, 12, n), P15_12, dev)
gk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
elt = ctx.galois_elt_from_step(1)
ev.rotate_vector_inplace(c, k, B, 1, {elt: gk}); ctx.synchronize()
ctx.profile_enable(True)
ev.rotate_vector_inplace(c, k, B, 1, {elt: gk}); ctx.synchronize()
prof = ctx.profile_fetch()
tot = sum(v["ms"] for v in prof.values())
print("rotate total ms", tot, {k_: round(v["ms"] / tot, 3) for k_, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})
