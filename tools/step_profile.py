"""Absolute kernel times (HIP events) of one step: tools/step_profile.py [batch] [cfg1|cfg3|cfg3sq|cfg3strict|cfg4|cfg4mul|cfg5]
cfg1 / cfg3 / cfg5: BFV multiply+relinearize at that config (cfg5: + mod_switch_to_next); cfg4: CKKS rotate_vector at
config 4; cfg4mul: CKKS multiply+relinearize. Run it under rocprofv3 --kernel-trace to see what the tags do not cover."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd")); sys.path.insert(0, ROOT)
import torch, bench, sealhip as S
from tools.bench_configs import mk, P15_12, P12, P16
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 1024
which = sys.argv[2] if len(sys.argv) > 2 else "cfg3"
n = 1 << 15
if which in ("cfg3", "cfg3sq", "cfg3strict"):
    k, pr = 7, bench.CFG3_PRIMES
    ctx = S.Context(S.SCHEME_BFV, 15, pr, 1, 786433, mode=S.MODE_STRICT if which == "cfg3strict" else S.MODE_PARITY)
elif which == "cfg1":
    n, k, pr = 1 << 12, 2, P12
    ctx = S.Context(S.SCHEME_BFV, 12, pr, 1, 786433)
elif which == "cfg5":
    n, k, pr = 1 << 16, 15, P16
    ctx = S.Context(S.SCHEME_BFV, 16, pr, 1, 786433)
else:
    k, pr = 11, P15_12
    ctx = S.Context(S.SCHEME_CKKS, 15, pr, 1, 0)
ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
ev = S.Evaluator(ctx)
x, y = mk(ctx, (B, 2, k, n), pr[:k], dev), mk(ctx, (B, 2, k, n), pr[:k], dev)
o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
o2 = torch.empty((B, 2, k - 1, n), dtype=torch.int64, device=dev)
rk = S.KSwitchKeys(ctx, mk(ctx, (k, 2, k + 1, n), pr, dev), n_digits=k, from_host=False)
elt = ctx.galois_elt_from_step(1) if which == "cfg4" else None
def step():
    if which == "cfg4":
        ev.rotate_vector_inplace(x, k, B, 1, {elt: rk})
    else:
        if which == "cfg3sq":
            ev.square(x, 2, k, B, o)
        else:
            ev.multiply(x, 2, y, 2, k, B, o)
        ev.relinearize_inplace(o, 3, k, B, [rk])
        if which == "cfg5":
            ev.mod_switch_to_next(o.view(B, 3, k, n)[:, :2].contiguous(), 2, k, B, o2)
step(); ctx.synchronize()
ctx.profile_enable(True)
for _ in range(3):
    step()
ctx.synchronize()
prof = ctx.profile_fetch()
print(which, {t: round(v["ms"] / 3, 3) for t, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}, "total", round(sum(v["ms"] for v in prof.values()) / 3, 2))
