"""Synthetic contexts / inputs shared by the oracle tests and the GPU parity tests
(generator spec: SURVEY.md Appendix B.2). Test infrastructure only."""
import ctypes as C

import numpy as np

import oracle_lib as O


def h(a):
    return "%016x" % O.fnv(a)


def end_to_end_inputs(row):
    """Key(s), operands a, b exactly as the survey's generator filled them (keys first)."""
    logn, scheme = row["logn"], row["scheme"]
    n = 1 << logn
    kmods = O.coeff_modulus_create(n, row["bits"])
    nsp = row["nsp"]
    nk = len(kmods)
    k = nk - nsp
    d = (k + nsp - 1) // nsp
    sm = O.SplitMix(0xC0FFEE + row["cfg"])

    def fill_key():
        return sm.fill(d * 2 * nk, n, kmods * (2 * d)).reshape(d, 2, nk, n)

    gk = fill_key() if scheme == 2 else None
    rk = fill_key()
    a = sm.fill(2 * k, n, kmods[:k] * 2).reshape(2, k, n)
    b = sm.fill(2 * k, n, kmods[:k] * 2).reshape(2, k, n)
    return dict(n=n, logn=logn, kmods=kmods, k=k, nk=nk, d=d, gk=gk, rk=rk, a=a, b=b)


def run_reference_chain(row):
    """SURVEY Appendix B.3 op chain through the oracle; returns {name: digest}."""
    L = O.lib()
    inp = end_to_end_inputs(row)
    n, k, logn = inp["n"], inp["k"], inp["logn"]
    scheme = row["scheme"]
    ctx = O.RefContext(scheme, logn, inp["kmods"], nsp=row["nsp"], t=row["t"])
    got = {}
    if scheme == 2:
        c = inp["a"].copy()
        elt = L.ref_galois_elt_from_step(n, 1, None)
        assert L.ref_apply_galois_inplace(C.byref(ctx.c), k, O.ptr(c), elt, O.ptr(inp["gk"])) == 0
        got["rotate"] = h(c)
    c = np.zeros((3, k, n), dtype=np.uint64)
    mul = L.ref_bfv_multiply if scheme == 1 else L.ref_ckks_multiply
    assert mul(C.byref(ctx.c), k, O.ptr(inp["a"]), 2, O.ptr(inp["b"]), 2, O.ptr(c)) == 0
    got["mul"] = h(c)
    keys = (C.c_void_p * 1)(inp["rk"].ctypes.data)
    assert L.ref_relinearize(C.byref(ctx.c), k, O.ptr(c), 3, keys) == 0
    c2 = c[:2].copy()
    got["relin"] = h(c2)
    o = np.zeros((2, k - 1, n), dtype=np.uint64)
    assert L.ref_mod_switch_scale_to_next(C.byref(ctx.c), k, O.ptr(c2), 2, O.ptr(o)) == 0
    got["modswitch" if scheme == 1 else "rescale"] = h(o)
    return got
