#!/usr/bin/env python3
"""Where the time of a *_host batch goes (config-3 shape, random words): the device op alone at the chunk size, then
multiply + relinearize through sealhip_evaluator_multiply_host on pageable and on registered buffers.
    SEALHIP_HOST_CHUNK=64 python tools/host_batch_probe.py [pairs, default 256]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd"))
import numpy as np
import sealhip as S

P = int(sys.argv[1]) if len(sys.argv) > 1 else 256
chunk = int(os.environ.get("SEALHIP_HOST_CHUNK", "64"))
PR = [36028797010444289, 36028797012606977, 36028797013000193, 36028797013327873, 36028797014376449,
      36028797014573057, 36028797014704129, 36028797017456641]
logn, n, k = 15, 1 << 15, 7
ctx = S.Context(S.SCHEME_BFV, logn, PR, 1, 786433)
ev = S.Evaluator(ctx)
rng = np.random.default_rng(0)


def rand(count, size, mods):
    return np.stack([rng.integers(0, p, size=(count, size, n), dtype=np.uint64) for p in mods], axis=2)


key = np.stack([rand(1, 2, PR)[0] for _ in range(k)])
rk = S.KSwitchKeys(ctx, key)
a, b = rand(chunk, 2, PR[:k]), rand(chunk, 2, PR[:k])
da, db = ctx.upload(a), ctx.upload(b)
d3 = ctx.alloc(chunk * 3 * k * n)
for _ in range(2):
    ev.multiply(da, 2, db, 2, k, chunk, d3)
    ev.relinearize_inplace(d3, 3, k, chunk, [rk])
ctx.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    ev.multiply(da, 2, db, 2, k, chunk, d3)
    ev.relinearize_inplace(d3, 3, k, chunk, [rk])
ctx.synchronize()
dt = (time.perf_counter() - t0) / 5
print("chunk %d: device op alone %.2f ms (%.0f ct/s)" % (chunk, dt * 1e3, chunk / dt))
words = 2 * k * n
blocks = [np.zeros((32, 2, k, n), dtype=np.uint64) for _ in range(3 * ((P + 31) // 32))]
nb = (P + 31) // 32
ha = [blocks[i // 32][i % 32] for i in range(P)]
hb = [blocks[nb + i // 32][i % 32] for i in range(P)]
ho = [blocks[2 * nb + i // 32][i % 32] for i in range(P)]
for i in range(P):
    ha[i][...] = a[i % chunk]
    hb[i][...] = b[i % chunk]
for label in ("pageable", "registered"):
    if label == "registered":
        t0 = time.perf_counter()
        for blk in blocks:
            ev.host_register(blk)
        print("registering %.2f GB: %.3f s" % (sum(x.nbytes for x in blocks) / 1e9, time.perf_counter() - t0))
    ev.multiply_host(ha, 2, hb, 2, k, ho, relin_keys=[rk])
    t0 = time.perf_counter()
    for _ in range(3):
        ev.multiply_host(ha, 2, hb, 2, k, ho, relin_keys=[rk])
    dt = (time.perf_counter() - t0) / 3
    print("chunk %d, %d pairs, %s: %.1f ms = %.0f ct/s, H2D %.1f GB/s" % (chunk, P, label, dt * 1e3, P / dt, P * 2 * words * 8 / dt / 1e9))
