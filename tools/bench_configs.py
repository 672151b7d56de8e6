#!/usr/bin/env python3
"""Secondary measurements for DESIGN.md (not the driver's bench line): the other BASELINE.json configs on one
MI355X and the PCIe-inclusive rate of config 3. Prints one JSON line per measurement."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd"))
sys.path.insert(0, ROOT)
import numpy as np
import torch

import bench
import sealhip as S

P14 = [1125899903107073, 1125899903500289, 1125899903795201, 1125899903827969, 1125899903991809, 1125899904679937]
P15_12 = [1125899885412353, 1125899885740033, 1125899886395393, 1125899887312897, 1125899896160257, 1125899899174913,
          1125899901665281, 1125899902124033, 1125899903107073, 1125899903500289, 1125899903827969, 1125899904679937]
P16 = [1125899864506369, 1125899865948161, 1125899870011393, 1125899870404609, 1125899877875713, 1125899879710721,
       1125899882987521, 1125899883380737, 1125899883642881, 1125899884036097, 1125899884167169, 1125899885740033,
       1125899886395393, 1125899887312897, 1125899902124033, 1125899903827969]
P12 = [68719230977, 68719403009, 137438822401]


def timed(fn, reps):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def mk(ctx, shape, moduli, dev):
    t = torch.empty(shape, dtype=torch.int64, device=dev)
    bench.fill_mod_rows(t, moduli)
    return t


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--only", default="")
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    torch.cuda.set_device(0)
    out = []

    def want(name):
        return not a.only or name in a.only.split(",")

    if want("cfg2"):
        logn, n = 14, 1 << 14
        ctx = S.Context(S.SCHEME_CKKS, logn, P14, 1, 0)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        P = 1024
        x = mk(ctx, (P, 6, n), P14, dev)  # key-level polynomials: 6 rows each (1024 x 6 = 6144 rows)
        # BASE_KEY at level k=5 = 5 ciphertext primes + 1 special prime = all 6 key primes
        fwd = timed(lambda: ctx.ntt_negacyclic_harvey(x, P, 5, S.BASE_KEY), 10)
        inv = timed(lambda: ctx.inverse_ntt_negacyclic_harvey(x, P, 5, S.BASE_KEY), 10)
        rows = P * 6
        out.append({"config": "cfg2 CKKS N=2^14 6 primes: batched NTT over 1024 polys (6144 rows)",
                    "forward_ntt_per_s": rows / fwd, "inverse_ntt_per_s": rows / inv,
                    "fwd_hbm_roofline_frac": rows * 16 * n / fwd / 8e12, "inv_hbm_roofline_frac": rows * 16 * n / inv / 8e12,
                    "fwd_plus_inv_ms": (fwd + inv) * 1e3})
    if want("cfg1"):
        logn, n = 12, 1 << 12
        ctx = S.Context(S.SCHEME_BFV, logn, P12, 1, 786433)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 4096, 2
        x, y = mk(ctx, (B, 2, k, n), P12[:k], dev), mk(ctx, (B, 2, k, n), P12[:k], dev)
        o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
        key = mk(ctx, (2, 2, 3, n), P12, dev)
        rk = S.KSwitchKeys(ctx, key, n_digits=2, from_host=False)

        def step():
            ev.multiply(x, 2, y, 2, k, B, o)
            ev.relinearize_inplace(o, 3, k, B, [rk])
        dt = timed(step, 5)
        out.append({"config": "cfg1 BFV N=4096 3 primes: multiply+relinearize, batch 4096", "ct_mul_relin_per_s": B / dt})
    if want("cfg4"):
        logn, n = 15, 1 << 15
        ctx = S.Context(S.SCHEME_CKKS, logn, P15_12, 1, 0)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 1024, 11  # 8192 ciphertexts / 8 GPUs
        c = mk(ctx, (B, 2, k, n), P15_12[:k], dev)
        key = mk(ctx, (k, 2, 12, n), P15_12, dev)
        gk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
        elt = ctx.galois_elt_from_step(1)
        dt = timed(lambda: ev.rotate_vector_inplace(c, k, B, 1, {elt: gk}), 3)
        # multiply + relinearize + rescale on the same context
        x, y = mk(ctx, (B, 2, k, n), P15_12[:k], dev), mk(ctx, (B, 2, k, n), P15_12[:k], dev)
        o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
        o2 = torch.empty((B, 2, k - 1, n), dtype=torch.int64, device=dev)

        def chain():
            ev.multiply(x, 2, y, 2, k, B, o)
            ev.relinearize_inplace(o, 3, k, B, [gk])
        dt2 = timed(chain, 3)
        out.append({"config": "cfg4 CKKS N=2^15 12 primes (k=11, 11 digits): per-GPU share 1024 ciphertexts",
                    "rotate_vector_per_s": B / dt, "ct_mul_relin_per_s": B / dt2})
    if want("cfg5"):
        logn, n = 16, 1 << 16
        ctx = S.Context(S.SCHEME_BFV, logn, P16, 1, 786433)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 256, 15
        x, y = mk(ctx, (B, 2, k, n), P16[:k], dev), mk(ctx, (B, 2, k, n), P16[:k], dev)
        o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
        o2 = torch.empty((B, 2, k - 1, n), dtype=torch.int64, device=dev)
        key = mk(ctx, (k, 2, 16, n), P16, dev)
        rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)

        def step():
            ev.multiply(x, 2, y, 2, k, B, o)
            ev.relinearize_inplace(o, 3, k, B, [rk])
            # relinearized ciphertext = first two polys of each item (batch stride stays 3 polys)
            ev.mod_switch_to_next(o.view(B, 3, k, n)[:, :2].contiguous(), 2, k, B, o2)
        dt = timed(step, 3)
        out.append({"config": "cfg5 BFV N=2^16 16 primes (k=15, |Bsk|=16, 15 digits): multiply+relinearize+mod_switch, batch 256",
                    "pipeline_per_s": B / dt})
    if want("f1"):
        # SURVEY 8(f1) rows at cfg3 size: streaming kernels, priced against the HBM roofline with their algorithmic bytes
        logn, n = 15, 1 << 15
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, 786433)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 1024, 7
        x, y = mk(ctx, (B, 2, k, n), pr[:k], dev), mk(ctx, (B, 2, k, n), pr[:k], dev)
        o = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
        rows = B * 2 * k
        dt_add = timed(lambda: ev.add(x, 2, y, 2, k, B, o), 10)
        dt_neg = timed(lambda: ev.negate(x, 2, k, B, o), 10)
        pn = mk(ctx, (k, n), pr[:k], dev)
        dt_mpn = timed(lambda: ev.multiply_plain_inplace(o, 2, k, B, pn, 0, ntt_form=True), 10)
        plain = torch.randint(0, 786433, (n,), dtype=torch.int64, device=dev)
        dt_mp = timed(lambda: ev.multiply_plain_inplace(o, 2, k, B, plain, 0, ntt_form=False), 5)
        dt_tr = timed(lambda: ctx.is_transparent(o, 2, k, B), 10)
        out.append({"config": "f1 rows at cfg3 size (BFV N=2^15, k=7), batch 1024 size-2 ciphertexts",
                    "add_ct_per_s": B / dt_add, "add_hbm_roofline_frac": rows * 24 * n / dt_add / 8e12,
                    "negate_ct_per_s": B / dt_neg, "negate_hbm_roofline_frac": rows * 16 * n / dt_neg / 8e12,
                    "multiply_plain_ntt_ct_per_s": B / dt_mpn, "multiply_plain_ntt_hbm_roofline_frac": rows * 16 * n / dt_mpn / 8e12,
                    "multiply_plain_ct_per_s": B / dt_mp,
                    "multiply_plain_ntt_equiv_frac": rows * 2 * 16 * n / dt_mp / 8e12,
                    "is_transparent_ct_per_s": B / dt_tr, "is_transparent_hbm_roofline_frac": (rows // 2) * 8 * n / dt_tr / 8e12})
    if want("f234"):
        # SURVEY 8(f2)/(f3)/(f4) rows at cfg3 size. Streaming + NTT work priced against the HBM roofline with the
        # bytes each operation has to move (inputs read once, outputs written once); the wire format is host-bound.
        logn, n, t = 15, 1 << 15, 786433
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, t)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 1024, 7
        a_ntt = mk(ctx, (B, k, n), pr[:k], dev)
        sk = mk(ctx, (8, n), pr, dev)
        e32 = torch.randint(-19, 20, (B, n), dtype=torch.int32, device=dev)
        ct = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
        dt_sym = timed(lambda: ctx.encrypt_zero_symmetric(k, False, a_ntt, e32, sk, B, ct), 5)
        pk = mk(ctx, (2, k, n), pr[:k], dev)
        u32 = torch.randint(-1, 2, (B, n), dtype=torch.int32, device=dev)
        e2 = torch.randint(-19, 20, (B, 2, n), dtype=torch.int32, device=dev)
        dt_asym = timed(lambda: ctx.encrypt_zero_asymmetric(k, False, pk, u32, e2, B, ct), 5)
        plain = torch.randint(0, t, (B, n), dtype=torch.int64, device=dev)
        dt_ap = timed(lambda: ev.add_plain_inplace(ct, 2, k, B, plain), 10)
        vals = torch.randint(0, t, (4 * B, n), dtype=torch.int64, device=dev)
        pl = torch.empty_like(vals)
        dt_enc = timed(lambda: ctx.batch_encode(vals, n, 4 * B, pl), 10)
        dt_dec = timed(lambda: ctx.batch_decode(pl, 4 * B, vals), 10)
        dt_val = timed(lambda: ctx.is_data_valid_for(ct, 2, k, B), 10)
        ctx.set_parms_id(k, (1, 2, 3, 4))
        info = S.CiphertextInfo()
        info.parms_id[:] = (1, 2, 3, 4)
        info.size, info.coeff_modulus_size, info.poly_modulus_degree, info.scale = 2, k, n, 1.0
        raw = ctx.save_ciphertext(info, ct[0])
        one = torch.empty((2, k, n), dtype=torch.int64, device=dev)
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.load_ciphertext(raw, one, capacity_words=one.numel())
        dt_load = (time.perf_counter() - t0) / 20
        t0 = time.perf_counter()
        for _ in range(20):
            ctx.save_ciphertext(info, one)
        dt_save = (time.perf_counter() - t0) / 20
        # CKKSEncoder at cfg4 size (N=2^15, k=11): 256 plaintexts of N/2 complex slots
        cctx = S.Context(S.SCHEME_CKKS, 15, P15_12, 1, 0)
        cctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        CB, ck = 256, 11
        cv = torch.randn((CB, n // 2, 2), dtype=torch.float64, device=dev) * 1000.0
        cpl = torch.empty((CB, ck, n), dtype=torch.int64, device=dev)
        cout = torch.empty((CB, n // 2, 2), dtype=torch.float64, device=dev)
        lib = S.lib()
        dt_cenc = timed(lambda: S._check(lib.sealhip_ckks_encode(cctx.handle, ck, cv.data_ptr(), n // 2, CB, 2.0 ** 40, cpl.data_ptr())), 5)
        dt_cdec = timed(lambda: S._check(lib.sealhip_ckks_decode(cctx.handle, ck, cpl.data_ptr(), CB, 2.0 ** 40, cout.data_ptr())), 5)
        row = 8 * n
        out.append({"config": "f2/f3/f4 rows at cfg3 size (BFV N=2^15, k=7, t=786433)",
                    # symmetric: read a (k rows) + sk, write c0, c1 (2k rows), inverse NTT of 2k rows (16N each)
                    "encrypt_zero_symmetric_ct_per_s": B / dt_sym,
                    "encrypt_zero_symmetric_hbm_frac": B * (3 * k * row + 2 * k * 2 * row) / dt_sym / 8e12,
                    # asymmetric: NTT(u) k rows, 2k products written, inverse NTT of 2k rows, + e
                    "encrypt_zero_asymmetric_ct_per_s": B / dt_asym,
                    "encrypt_zero_asymmetric_hbm_frac": B * (k * 2 * row + 2 * k * 2 * row + 2 * k * 2 * row + 2 * k * 2 * row) / dt_asym / 8e12,
                    "add_plain_ct_per_s": B / dt_ap, "add_plain_hbm_frac": B * (row + 2 * k * row) / dt_ap / 8e12,
                    "batch_encode_plain_per_s": 4 * B / dt_enc, "batch_encode_hbm_frac": 4 * B * 4 * row / dt_enc / 8e12,
                    "batch_decode_plain_per_s": 4 * B / dt_dec, "batch_decode_hbm_frac": 4 * B * 6 * row / dt_dec / 8e12,
                    "is_data_valid_for_ct_per_s": B / dt_val, "is_data_valid_for_hbm_frac": B * 2 * k * row / dt_val / 8e12,
                    "wire_load_GBps_pageable_host": len(raw) / dt_load / 1e9, "wire_save_GBps_pageable_host": len(raw) / dt_save / 1e9,
                    "wire_bytes_per_ct": len(raw),
                    # CKKS encode: read N/2 complex, FFT over N complex (32N B per pass), write k rows + NTT of k rows
                    "ckks_encode_plain_per_s_cfg4": CB / dt_cenc, "ckks_encode_hbm_frac": CB * (8 * n + 32 * n + ck * 3 * 8 * n) / dt_cenc / 8e12,
                    "ckks_decode_plain_per_s_cfg4": CB / dt_cdec, "ckks_decode_hbm_frac": CB * (ck * 3 * 8 * n + 32 * n + 8 * n) / dt_cdec / 8e12})
    if want("pcie"):
        logn, n = 15, 1 << 15
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, 786433)
        ctx.use_default_stream()  # torch fills run on the legacy default stream: same stream, ordered
        ev = S.Evaluator(ctx)
        B, k = 256, 7
        ha = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        hb = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        ho = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        for t in (ha, hb):
            for i, p in enumerate(pr[:k]):
                t[:, :, i, :] = torch.randint(0, p, (B, 2, n), dtype=torch.int64)
        key = mk(ctx, (k, 2, 8, n), pr, dev)
        rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
        da = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
        db = torch.empty_like(da)
        o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)

        def step():
            da.copy_(ha, non_blocking=True)
            db.copy_(hb, non_blocking=True)
            ev.multiply(da, 2, db, 2, k, B, o)
            ev.relinearize_inplace(o, 3, k, B, [rk])
            ho.copy_(o[:, :2], non_blocking=True)
        dt = timed(step, 3)
        out.append({"config": "cfg3 with PCIe: pinned host -> device (2 x 3.67 MB/ct), multiply+relinearize, device -> host (3.67 MB/ct), batch 256, one stream (no overlap)",
                    "ct_mul_relin_per_s_pcie_inclusive": B / dt})
    if want("latency"):
        # small-batch behaviour of cfg3: time of one multiply+relinearize call for 1..64 ciphertext pairs (launch/latency bound)
        logn, n = 15, 1 << 15
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, 786433)
        s_lat = torch.cuda.Stream()  # graphs cannot be captured from the legacy default stream: an explicit one
        ctx.set_stream(s_lat.cuda_stream)
        ev = S.Evaluator(ctx)
        k = 7
        key = mk(ctx, (k, 2, 8, n), pr, dev)
        torch.cuda.synchronize()
        rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
        res = {}
        for B in (1, 2, 4, 8, 16, 64, 256):
            x, y = mk(ctx, (B, 2, k, n), pr[:k], dev), mk(ctx, (B, 2, k, n), pr[:k], dev)
            o = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
            torch.cuda.synchronize()

            def step():
                ev.multiply(x, 2, y, 2, k, B, o)
                ev.relinearize_inplace(o, 3, k, B, [rk])
            dt = timed(step, 20)
            res["batch_%d" % B] = {"ms_per_call": dt * 1e3, "ct_per_s": B / dt}
            if B <= 16:  # the same call sequence replayed as one hipGraph launch
                g = ctx.capture(step)
                dtg = timed(g.launch, 50)
                res["batch_%d" % B].update({"ms_per_call_hipgraph": dtg * 1e3, "ct_per_s_hipgraph": B / dtg})
                del g
        out.append({"config": "cfg3 small batches: one multiply+relinearize call, device-resident", **res})
    if want("pcie_overlap"):
        # the same boundary with the copies on their own HIP streams: chunks of 64 ciphertext pairs, double-buffered
        # device staging, H2D(c+1) and D2H(c-1) run while chunk c computes
        logn, n = 15, 1 << 15
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, 786433)
        s_comp, s_in, s_out = torch.cuda.Stream(), torch.cuda.Stream(), torch.cuda.Stream()
        ctx.set_stream(s_comp.cuda_stream)
        ev = S.Evaluator(ctx)
        B, k, CH = 512, 7, 64
        ha = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        hb = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        ho = torch.empty((B, 2, k, n), dtype=torch.int64).pin_memory()
        for t in (ha, hb):
            for i, p in enumerate(pr[:k]):
                t[:, :, i, :] = torch.randint(0, p, (B, 2, n), dtype=torch.int64)
        key = mk(ctx, (k, 2, 8, n), pr, dev)
        rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
        da = [torch.empty((CH, 2, k, n), dtype=torch.int64, device=dev) for _ in range(2)]
        db = [torch.empty((CH, 2, k, n), dtype=torch.int64, device=dev) for _ in range(2)]
        do = [torch.empty((CH, 3, k, n), dtype=torch.int64, device=dev) for _ in range(2)]
        res = [torch.empty((CH, 2, k, n), dtype=torch.int64, device=dev) for _ in range(2)]
        nchunks = B // CH

        def run():
            in_ready = [torch.cuda.Event() for _ in range(nchunks)]
            comp_done = [torch.cuda.Event() for _ in range(nchunks)]
            out_done = [torch.cuda.Event() for _ in range(nchunks)]
            for c in range(nchunks):
                sl, bi = slice(c * CH, (c + 1) * CH), c & 1
                with torch.cuda.stream(s_in):
                    if c >= 2:
                        s_in.wait_event(comp_done[c - 2])  # the staging buffers of chunk c-2 are free again
                    da[bi].copy_(ha[sl], non_blocking=True)
                    db[bi].copy_(hb[sl], non_blocking=True)
                    in_ready[c].record(s_in)
                with torch.cuda.stream(s_comp):
                    s_comp.wait_event(in_ready[c])
                    if c >= 2:
                        s_comp.wait_event(out_done[c - 2])  # res[bi] has left for the host
                    ev.multiply(da[bi], 2, db[bi], 2, k, CH, do[bi])
                    ev.relinearize_inplace(do[bi], 3, k, CH, [rk])
                    ev.resize(do[bi], 3, 2, k, CH, res[bi])
                    comp_done[c].record(s_comp)
                with torch.cuda.stream(s_out):
                    s_out.wait_event(comp_done[c])
                    ho[sl].copy_(res[bi], non_blocking=True)
                    out_done[c].record(s_out)
            torch.cuda.synchronize()

        run()
        t0 = time.perf_counter()
        for _ in range(3):
            run()
        dt = (time.perf_counter() - t0) / 3
        # correctness of the overlapped pipeline against the plain one on the first chunk
        chk = torch.empty((CH, 3, k, n), dtype=torch.int64, device=dev)
        x, y = ha[:CH].to(dev), hb[:CH].to(dev)
        ev.multiply(x, 2, y, 2, k, CH, chk)
        ev.relinearize_inplace(chk, 3, k, CH, [rk])
        ctx.synchronize()
        ok = bool(torch.equal(chk[:, :2].cpu(), ho[:CH]))
        out.append({"config": "cfg3 with PCIe, copies overlapped: 3 HIP streams (H2D / compute / D2H), chunks of 64 pairs, double-buffered staging, batch 512",
                    "ct_mul_relin_per_s_pcie_inclusive": B / dt, "matches_unpipelined": ok,
                    "h2d_GBps": B * 2 * 2 * k * n * 8 / dt / 1e9, "d2h_GBps": B * 2 * k * n * 8 / dt / 1e9})
    if want("host_batch"):
        # the library's own boundary for the reference's objects: 512 SEPARATELY ALLOCATED pageable host ciphertext pairs
        # (what a vector<seal::Ciphertext> is) through sealhip_evaluator_multiply_host with relinearization -- gather threads,
        # pinned double-buffered staging, three streams, all inside the library
        import numpy as np
        logn, n = 15, 1 << 15
        pr = bench.CFG3_PRIMES
        ctx = S.Context(S.SCHEME_BFV, logn, pr, 1, 786433)
        ev = S.Evaluator(ctx)
        B, k = 512, 7
        rng = np.random.default_rng(5)
        key = mk(ctx, (k, 2, 8, n), pr, dev)
        torch.cuda.synchronize()
        rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)

        def one():
            return np.stack([np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in pr[:k]]) for _ in range(2)])
        base_a, base_b = one(), one()
        ha = [base_a.copy() for _ in range(B)]
        hb = [base_b.copy() for _ in range(B)]
        ho = [np.zeros((2, k, n), dtype=np.uint64) for _ in range(B)]
        ev.multiply_host(ha, 2, hb, 2, k, ho, relin_keys=[rk])  # staging buffers, arena
        t0 = time.perf_counter()
        reps = 3
        for _ in range(reps):
            ev.multiply_host(ha, 2, hb, 2, k, ho, relin_keys=[rk])
        dt = (time.perf_counter() - t0) / reps
        x, y = ctx.upload(base_a[None]), ctx.upload(base_b[None])
        chk = ctx.alloc(3 * k * n)
        ev.multiply(x, 2, y, 2, k, 1, chk)
        ev.relinearize_inplace(chk, 3, k, 1, [rk])
        ok = bool(np.array_equal(chk.download((3, k, n))[:2], ho[0]) and np.array_equal(ho[0], ho[-1]))
        out.append({"config": "cfg3 through sealhip_evaluator_multiply_host (+relinearize): 512 separately allocated pageable host "
                              "ciphertext pairs, library-side gather threads + pinned double-buffered staging + 3 streams",
                    "ct_mul_relin_per_s_pcie_inclusive": B / dt, "matches_device_path": ok,
                    "h2d_GBps": B * 2 * 2 * k * n * 8 / dt / 1e9, "d2h_GBps": B * 2 * k * n * 8 / dt / 1e9})
    for line in out:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
