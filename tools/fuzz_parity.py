#!/usr/bin/env python3
"""Randomised differential run of the HIP path against the CPU oracle (test infrastructure): random ring sizes, prime
counts and sizes, special-prime counts, schemes, batch sizes; multiply -> relinearize -> mod_switch/rescale -> apply_galois,
plus add/sub and multiply_plain. Usage: python tools/fuzz_parity.py [iterations] [seed]"""
import ctypes as C
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd"))
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np

import oracle_lib as O
import sealhip as S

L = O.lib()


def rand_ct(rng, mods, size, n, count):
    return np.stack([rng.integers(0, p, size=(count, size, n), dtype=np.uint64) for p in mods], axis=2).copy()


def one(rng, it):
    scheme = int(rng.integers(1, 3))
    logn = int(rng.choice([3, 4, 5, 8, 10, 11, 12, 13, 14, 15, 16]))
    n = 1 << logn
    nsp = int(rng.integers(1, 4))
    k = int(rng.integers(2, 7))
    bits = [int(rng.integers(max(25, logn + 4), 60)) for _ in range(k + nsp)]
    try:
        kmods = O.coeff_modulus_create(n, bits)
    except Exception:
        return "skip"
    if len(set(kmods)) != len(kmods):
        return "skip"
    t = 65537 if scheme == 1 else 0
    count = int(rng.choice([1, 2, 5, 17, 33])) if logn <= 12 else (int(rng.choice([1, 3, 17])) if logn <= 14 else int(rng.choice([1, 2])))
    ctx = S.Context(scheme, logn, kmods, nsp, t)
    ev = S.Evaluator(ctx)
    ref = O.RefContext(scheme, logn, kmods, nsp=nsp, t=t)
    d = (k + nsp - 1) // nsp
    key = np.stack([rand_ct(rng, kmods, 2, n, 1)[0] for _ in range(d)])
    dkey = S.KSwitchKeys(ctx, key)
    a, b = rand_ct(rng, kmods[:k], 2, n, count), rand_ct(rng, kmods[:k], 2, n, count)
    out = ctx.alloc(count * 3 * k * n)
    ev.multiply(ctx.upload(a), 2, ctx.upload(b), 2, k, count, out)
    got = out.download((count, 3, k, n))
    mul = L.ref_bfv_multiply if scheme == 1 else L.ref_ckks_multiply
    keys = (C.c_void_p * 1)(key.ctypes.data)
    ev.relinearize_inplace(out, 3, k, count, [dkey])
    got_r = out.download((count, 3, k, n))
    c2 = got_r[:, :2].copy()
    low = ctx.alloc(count * 2 * (k - 1) * n)
    (ev.mod_switch_to_next if scheme == 1 else ev.rescale_to_next)(ctx.upload(c2), 2, k, count, low)
    got_l = low.download((count, 2, k - 1, n))
    steps = int(rng.integers(1, max(2, n // 2)))
    elt = ctx.galois_elt_from_step(steps)
    g = ctx.upload(c2)
    ev.apply_galois_inplace(g, k, count, elt, dkey)
    got_g = g.download((count, 2, k, n))
    for i in range(count):
        exp = np.zeros((3, k, n), dtype=np.uint64)
        assert mul(C.byref(ref.c), k, O.ptr(a[i]), 2, O.ptr(b[i]), 2, O.ptr(exp)) == 0
        assert np.array_equal(got[i], exp), ("multiply", it, i)
        assert L.ref_relinearize(C.byref(ref.c), k, O.ptr(exp), 3, keys) == 0
        assert np.array_equal(got_r[i, :2], exp[:2]), ("relinearize", it, i)
        e2 = exp[:2].copy()
        lo = np.zeros((2, k - 1, n), dtype=np.uint64)
        assert L.ref_mod_switch_scale_to_next(C.byref(ref.c), k, O.ptr(e2), 2, O.ptr(lo)) == 0
        assert np.array_equal(got_l[i], lo), ("mod_switch", it, i)
        assert L.ref_apply_galois_inplace(C.byref(ref.c), k, O.ptr(e2), elt, O.ptr(key)) == 0
        assert np.array_equal(got_g[i], e2), ("galois", it, i)
    return "%s logn=%d k=%d nsp=%d count=%d bits=%s" % ("BFV" if scheme == 1 else "CKKS", logn, k, nsp, count, bits)


def main():
    iters = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    seed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
    rng = np.random.default_rng(seed)
    done = 0
    for it in range(iters):
        r = one(rng, it)
        if r != "skip":
            done += 1
            print("ok", it, r, flush=True)
    print("fuzz: %d cases bit-exact" % done)


if __name__ == "__main__":
    main()
