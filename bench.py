#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sealhip engine.

Metric (BASELINE.json): ciphertext multiply+relinearize/s (and forward-NTT/s) at N=2^15, 8 RNS primes.
A step = Evaluator::multiply + Evaluator::relinearize over one batch of independent synthetic BFV
ciphertexts that are already resident in HBM (config 3 of BASELINE.json: N=2^15, {55}x8 primes, k=7,
|Bsk|=8, 7 key digits, t=786433, PARITY mode = bit-exact with the reference).

    python bench.py --gpus N --steps K --warmup W [--batch B]

For N > 1 the driver launches one rank per GPU (torch.distributed, backend nccl = RCCL); the batch of
independent ciphertexts is sharded contiguously over ranks with NO data-path collective (weak scaling:
every rank owns `--batch` ciphertexts); the only collective besides the timing barrier is the gather of
the 8-byte result digests to rank 0 after the timed region.

Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
for _p in (os.path.join(ROOT, "gemini-seal_amd"), os.path.join(ROOT, "tests")):
    if _p not in sys.path:
        sys.path.insert(0, _p)

import numpy as np  # noqa: E402
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

HBM_PEAK_GBS = 8000.0  # MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec
LOGN, BITS, NSP, PLAIN_T = 15, [55] * 8, 1, 786433
# CoeffModulus::Create(32768, {55}x8); recomputed by the engine-independent helper below when the
# oracle is available, hard-wired here so that the timed path never touches oracle/.
CFG3_PRIMES = [36028797010444289, 36028797012606977, 36028797013000193, 36028797013327873, 36028797014376449,
               36028797014573057, 36028797014704129, 36028797017456641]


# ------------------------------------------------------------------ multi-rank helpers (covered by a gloo test)
def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` independent ciphertexts for `rank`."""
    return (rank * total) // world, ((rank + 1) * total) // world


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def max_over_ranks(seconds):
    if not _dist_on():
        return seconds
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    t = torch.tensor([seconds], dtype=torch.float64, device=dev)
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def gather_digests(digest):
    """The final gather: 8-byte digests of every rank's output slice to rank 0."""
    if not _dist_on():
        return [digest]
    dev = "cuda" if dist.get_backend() == "nccl" else "cpu"
    mine = torch.tensor([digest], dtype=torch.int64, device=dev)
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t.item()) for t in out]


def barrier_sync():
    if torch.cuda.is_available():
        torch.cuda.synchronize()
    if _dist_on():
        dist.barrier()
    if torch.cuda.is_available():
        torch.cuda.synchronize()


# ------------------------------------------------------------------ synthetic data
def fill_mod_rows(t, moduli):
    """t[..., i, :] uniform in [0, moduli[i]) -- the distribution real ciphertexts/keys have."""
    for i, p in enumerate(moduli):
        view = t[..., i, :]
        view.copy_(torch.randint(0, p, view.shape, dtype=torch.int64, device=t.device))


def cheap_digest(t):
    """Order-dependent 63-bit digest computed on the device (for the cross-rank gather)."""
    flat = t.reshape(-1)
    idx = torch.arange(flat.numel(), device=flat.device, dtype=torch.int64)
    return int(((flat * 0x9E3779B97F4A7C15 + idx) ^ (flat >> 29)).sum().item() & 0x7FFFFFFFFFFFFFFF)


# ------------------------------------------------------------------ CPU baseline (oracle = "port")
def cpu_baseline(threads, per_thread):
    """Times the CPU oracle (oracle/sealref.c, digest-identical to the reference) on a bounded sample of the
    SAME workload: `threads` worker threads, each doing `per_thread` multiply+relinearize on its own
    ciphertexts (the reference is single-threaded per call and thread-safe across calls)."""
    from concurrent.futures import ThreadPoolExecutor

    import oracle_lib as O

    L = O.lib()
    n = 1 << LOGN
    kmods = O.coeff_modulus_create(n, BITS)
    assert kmods == CFG3_PRIMES
    ref = O.RefContext(1, LOGN, kmods, nsp=NSP, t=PLAIN_T)
    k = ref.k_first
    ref.rns_tool(k)  # build shared constants before the threads start
    rng = np.random.default_rng(1)

    def rows(mods):
        return np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in mods])

    key = np.stack([rows(kmods * 2).reshape(2, len(kmods), n) for _ in range(k)])
    work = [(rows(kmods[:k] * 2), rows(kmods[:k] * 2), np.zeros((3, k, n), dtype=np.uint64)) for _ in range(threads)]
    keys = (C.c_void_p * 1)(key.ctypes.data)

    def run(item):
        a, b, out = item
        for _ in range(per_thread):
            assert L.ref_bfv_multiply(C.byref(ref.c), k, O.ptr(a), 2, O.ptr(b), 2, O.ptr(out)) == 0
            assert L.ref_relinearize(C.byref(ref.c), k, O.ptr(out), 3, keys) == 0

    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, work))
    dt = time.perf_counter() - t0
    # the same on one thread (SURVEY 8d asks for both)
    t1 = time.perf_counter()
    run((work[0][0], work[0][1], work[0][2]))
    one_thread = per_thread / (time.perf_counter() - t1)
    # forward NTT/s on one core, same primes
    x = rows(kmods[:k])
    t1 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        for i in range(k):
            L.ref_ntt_forward(O.ptr(x[i]), ref.tables(i), 0)
    ntt_s = reps * k / (time.perf_counter() - t1)
    return {
        "value": threads * per_thread / dt,
        "unit": "ct_mul_relin/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d threads x %d BFV multiply+relinearize at N=2^15, 8 primes (same workload, %d ciphertexts)"
                  % (threads, per_thread, threads * per_thread),
        "seconds": dt,
        "value_1thread": one_thread,
        "forward_ntt_per_s_1core": ntt_s,
    }


# ------------------------------------------------------------------ main
def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SEALHIP_BENCH_BATCH", "4096")),
                    help="independent ciphertext pairs per GPU (BASELINE config 3: 4096)")
    ap.add_argument("--ntt-polys", type=int, default=4096,
                    help="polynomials (x7 rows) in the NTT-only section; 4096 = the batch of the step (the rate grows "
                         "with the launch: 31 %% of the roofline at 7 k rows, 36-38 %% at 29-57 k, DESIGN.md section 6)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    args = ap.parse_args()

    import sealhip as S

    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if not torch.cuda.is_available() or S.num_devices() < 1:
        raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
    torch.cuda.set_device(local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    assert args.gpus == world, "--gpus must equal the number of launched ranks"
    dev = torch.device("cuda", local_rank)

    n = 1 << LOGN
    kmods = CFG3_PRIMES
    ctx = S.Context(S.SCHEME_BFV, LOGN, kmods, NSP, PLAIN_T, device=local_rank)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)  # everything on torch's current stream
    ev = S.Evaluator(ctx)
    k, nk = ctx.k_first, len(kmods)
    B = args.batch

    torch.manual_seed(1234 + rank)
    a = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
    b = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
    out = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
    key = torch.empty((k, 2, nk, n), dtype=torch.int64, device=dev)
    fill_mod_rows(a, kmods[:k])
    fill_mod_rows(b, kmods[:k])
    torch.manual_seed(99)  # the relinearisation key is replicated on every GPU
    fill_mod_rows(key, kmods)
    rk = S.KSwitchKeys(ctx, key, n_digits=k, from_host=False)
    del key

    def step():
        ev.multiply(a, 2, b, 2, k, B, out)
        ev.relinearize_inplace(out, 3, k, B, [rk])

    for _ in range(args.warmup):
        step()
    barrier_sync()
    ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier_sync()
    dt = time.perf_counter() - t0
    prof = ctx.profile_fetch()
    ctx.profile_enable(False)
    dt = max_over_ranks(dt)
    value = world * B * args.steps / dt

    # ---- roofline of the dominant kernel, from HIP events recorded on the launch stream during the timed steps
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    dominant = max(prof, key=lambda tkey: prof[tkey]["ms"])
    shares = {tkey: round(v["ms"] / total_ms, 4) for tkey, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # launches per full row transform at N=2^15: the single-pass forward kernel completes a transform per launch,
    # the tiled pass kernel needs two launches (each then counts for half of the 16*N algorithmic bytes)
    PASSES = {"ntt_fwd_half": 1, "ntt_fwd_pass": 2, "ntt_inv_pass": 2}

    def ntt_roofline(tag):
        v = prof.get(tag)
        if not v or not v["launches"]:
            return None
        rows_per_launch = v["units"] / v["launches"]
        avg_s = v["ms"] / v["launches"] / 1e3
        # 16*N bytes per row transform
        alg_bytes = rows_per_launch * 16 * n / PASSES[tag]
        ach = alg_bytes / avg_s / 1e9
        return {"kernel": tag, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_s * 1e3,
                "rows_per_launch": rows_per_launch, "algorithmic_bytes_per_launch": alg_bytes,
                "launches": v["launches"], "share_of_step_kernel_time": v["ms"] / total_ms}

    roof = ntt_roofline(dominant) if dominant in PASSES else (ntt_roofline("ntt_fwd_half") or ntt_roofline("ntt_fwd_pass"))
    if roof is None:
        roof = {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": None, "traffic": None}
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if os.path.exists(tpath) and roof.get("rows_per_launch"):
        # HBM bytes per RNS row per launch, measured with separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes
        # (tools/summarize_prof.py, gfx950 correction applied), scaled to this run's rows per launch
        try:
            per_row = json.load(open(tpath))[roof["kernel"]]["hbm_bytes_per_row_per_launch"]
            roof["traffic"] = per_row * roof["rows_per_launch"]
        except Exception:
            pass

    # ---- NTT-only section: forward-NTT/s (the other half of the BASELINE metric), same primes, same device
    P = args.ntt_polys
    x = torch.empty((P, k, n), dtype=torch.int64, device=dev)
    fill_mod_rows(x, kmods[:k])
    ctx.ntt_negacyclic_harvey(x, P, k)
    barrier_sync()
    ctx.profile_enable(True)
    reps = 10
    t1 = time.perf_counter()
    for _ in range(reps):
        ctx.ntt_negacyclic_harvey(x, P, k)
    barrier_sync()
    ntt_dt = max_over_ranks(time.perf_counter() - t1)
    nprof = ctx.profile_fetch()
    ctx.profile_enable(False)
    ntt_rows = P * k * reps
    ntt_kernel_s = sum(v["ms"] for tkey, v in nprof.items() if tkey.startswith("ntt_fwd")) / 1e3
    ntt = {
        "forward_ntt_per_s": world * ntt_rows / ntt_dt,
        "forward_ntt_per_s_kernel_time": ntt_rows / ntt_kernel_s,
        "hbm_roofline_frac": (ntt_rows * 16 * n / ntt_kernel_s) / 1e9 / HBM_PEAK_GBS,
        "rows": P * k, "reps": reps,
    }

    digests = gather_digests(cheap_digest(out[: min(B, 4)]))

    cpu = None
    if rank == 0 and world == 1 and not args.no_cpu_baseline:
        threads = max(1, min(16, os.cpu_count() or 1))
        cpu = cpu_baseline(threads, 6)
        cpu["gpu_over_cpu_allcore"] = value / cpu["value"]

    if rank == 0:
        line = {
            "metric": "ciphertext multiply+relinearize/s (BFV N=2^15, 8 primes, bit-exact PARITY mode)",
            "value": value,
            "unit": "ct_mul_relin/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": "BASELINE config 3: BFV N=2^15, {55}x8 primes (k=7, |Bsk|=8, 7 digits), "
                                   "Evaluator::multiply + relinearize over independent ciphertexts resident in HBM",
                       "ciphertexts_per_gpu": B, "global_batch": world * B, "mode": "PARITY",
                       "parallelism": "dp%d (independent ciphertexts sharded, no data-path collective)" % world},
            "roofline": roof,
            "cpu_baseline": cpu,
            "ntt": ntt,
            "kernel_time_shares": shares,
            "rank_digests": ["%016x" % d for d in digests],
        }
        print(json.dumps(line))
    if _dist_on():
        dist.barrier()
        dist.destroy_process_group()


if __name__ == "__main__":
    main()
