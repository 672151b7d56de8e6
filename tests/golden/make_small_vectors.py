#!/usr/bin/env python3
"""Generates tests/golden/small_vectors.json: full small-N input/output vectors from the CPU oracle AFTER it has
been pinned by reference_kats.json and survey_digests.json (run tests/test_oracle.py first). They let the GPU box
check the HIP path (and the oracle build it received) against committed numbers, word for word."""
import ctypes as C
import json
import os
import sys

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
import oracle_lib as O  # noqa: E402

L = O.lib()


def rows(rng, mods, n):
    return np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in mods])


def chain(scheme, logn, bits, nsp, t, seed):
    n = 1 << logn
    kmods = O.coeff_modulus_create(n, bits)
    nk = len(kmods)
    k = nk - nsp
    d = (k + nsp - 1) // nsp
    rng = np.random.default_rng(seed)
    key = np.stack([rows(rng, kmods * 2, n).reshape(2, nk, n) for _ in range(d)])
    a = rows(rng, kmods[:k] * 2, n).reshape(2, k, n)
    b = rows(rng, kmods[:k] * 2, n).reshape(2, k, n)
    ctx = O.RefContext(scheme, logn, kmods, nsp=nsp, t=t)
    out = {"scheme": scheme, "logn": logn, "key_moduli": kmods, "nsp": nsp, "t": t, "k": k,
           "key": key.tolist(), "a": a.tolist(), "b": b.tolist()}
    c = np.zeros((3, k, n), dtype=np.uint64)
    mul = L.ref_bfv_multiply if scheme == 1 else L.ref_ckks_multiply
    assert mul(C.byref(ctx.c), k, O.ptr(a), 2, O.ptr(b), 2, O.ptr(c)) == 0
    out["multiply"] = c.tolist()
    keys = (C.c_void_p * 1)(key.ctypes.data)
    assert L.ref_relinearize(C.byref(ctx.c), k, O.ptr(c), 3, keys) == 0
    c2 = c[:2].copy()
    out["relinearize"] = c2.tolist()
    o = np.zeros((2, k - 1, n), dtype=np.uint64)
    assert L.ref_mod_switch_scale_to_next(C.byref(ctx.c), k, O.ptr(c2), 2, O.ptr(o)) == 0
    out["mod_switch_scale_to_next"] = o.tolist()
    elt = L.ref_galois_elt_from_step(n, 3, None)
    g = c2.copy()
    assert L.ref_apply_galois_inplace(C.byref(ctx.c), k, O.ptr(g), elt, O.ptr(key)) == 0
    out["galois_elt_step3"] = int(elt)
    out["apply_galois"] = g.tolist()
    return out


def ntt_case(logn, p, seed):
    n = 1 << logn
    rng = np.random.default_rng(seed)
    x = rng.integers(0, p, size=n, dtype=np.uint64)
    t = O.Tables(logn, p)
    res = {"logn": logn, "p": p, "x": x.tolist()}
    for name, fn in (("fwd_lazy", lambda r: L.ref_ntt_forward_lazy(O.ptr(r), C.byref(t.t), 0)),
                     ("fwd", lambda r: L.ref_ntt_forward(O.ptr(r), C.byref(t.t), 0)),
                     ("inv_lazy", lambda r: L.ref_ntt_inverse_lazy(O.ptr(r), C.byref(t.t))),
                     ("inv", lambda r: L.ref_ntt_inverse(O.ptr(r), C.byref(t.t)))):
        r = x.copy()
        fn(r)
        res[name] = r.tolist()
    return res


def main():
    data = {
        "_generator": "tests/golden/make_small_vectors.py (oracle output; the oracle is pinned by the other two files)",
        "ntt": [ntt_case(3, O.get_primes(8, 30, 1)[0], 1), ntt_case(5, O.get_primes(32, 59, 1)[0], 2),
                ntt_case(6, O.get_primes(64, 60, 1)[0], 3)],
        "chains": [chain(1, 5, [30, 30, 30, 30, 31], 2, 257, 11), chain(2, 5, [40, 40, 40, 41], 1, 0, 12)],
    }
    with open(os.path.join(HERE, "small_vectors.json"), "w") as f:
        json.dump(data, f)
    print("wrote", os.path.getsize(os.path.join(HERE, "small_vectors.json")), "bytes")


if __name__ == "__main__":
    main()
