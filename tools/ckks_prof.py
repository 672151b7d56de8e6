"""Kernel-time split of CKKSEncoder encode / decode at cfg4 size (N=2^15, k=11), from the engine's HIP-event profiler."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd")); sys.path.insert(0, ROOT)
import torch, sealhip as S
from tools.bench_configs import P15_12
dev = torch.device("cuda", 0)
n, k, B = 1 << 15, 11, 256
ctx = S.Context(S.SCHEME_CKKS, 15, P15_12, 1, 0)
ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
v = torch.randn((B, n // 2, 2), dtype=torch.float64, device=dev) * 1000
pl = torch.empty((B, k, n), dtype=torch.int64, device=dev)
out = torch.empty((B, n // 2, 2), dtype=torch.float64, device=dev)
L = S.lib()
enc = lambda: S._check(L.sealhip_ckks_encode(ctx.handle, k, v.data_ptr(), n // 2, B, 2.0 ** 40, pl.data_ptr()))
dec = lambda: S._check(L.sealhip_ckks_decode(ctx.handle, k, pl.data_ptr(), B, 2.0 ** 40, out.data_ptr()))
for name, fn in (("encode", enc), ("decode", dec)):
    fn(); ctx.synchronize()
    ctx.profile_enable(True); fn(); ctx.synchronize()
    prof = ctx.profile_fetch(); ctx.profile_enable(False)
    tot = sum(x["ms"] for x in prof.values())
    print(name, "kernel ms", round(tot, 3), {t: round(x["ms"], 3) for t, x in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])})
