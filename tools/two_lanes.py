"""Does running independent halves of a batch from several host threads (one engine lane = stream + arena each) overlap the
memory-bound kernels of one with the instruction-bound ones of the other? tools/two_lanes.py [batch] [threads...]"""
import os, sys, threading, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd")); sys.path.insert(0, ROOT)
import torch, bench, sealhip as S
from tools.bench_configs import mk
dev = torch.device("cuda", 0)
B = int(sys.argv[1]) if len(sys.argv) > 1 else 2048
threads = [int(v) for v in sys.argv[2:]] or [1, 2, 3]
n, k, pr = 1 << 15, 7, bench.CFG3_PRIMES
ctx = S.Context(S.SCHEME_BFV, 15, pr, 1, 786433)
ev = S.Evaluator(ctx)
x, y = mk(ctx, (B, 2, k, n), pr[:k], dev), mk(ctx, (B, 2, k, n), pr[:k], dev)
o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
rk = S.KSwitchKeys(ctx, mk(ctx, (k, 2, k + 1, n), pr, dev), n_digits=k, from_host=False)
torch.cuda.synchronize()
def work(lo, hi):
    m = hi - lo
    ev.multiply(x[lo:hi], 2, y[lo:hi], 2, k, m, o[lo:hi])
    ev.relinearize_inplace(o[lo:hi], 3, k, m, [rk])
    ctx.synchronize()
def run(T):
    ths = [threading.Thread(target=work, args=(i * B // T, (i + 1) * B // T)) for i in range(T)]
    t0 = time.perf_counter()
    for t in ths: t.start()
    for t in ths: t.join()
    torch.cuda.synchronize()
    return time.perf_counter() - t0
ref = None
for T in threads:
    run(T)
    dt = min(run(T) for _ in range(3))
    dig = int(o.view(-1)[::997].sum().item())
    ref = dig if ref is None else ref
    print({"threads": T, "ct_per_s": round(B / dt), "ms": round(dt * 1e3, 2), "same_result": dig == ref})
