#!/usr/bin/env python3
"""SURVEY 8(e) latency mode: ONE key switch (rotate / relinearize of a single ciphertext batch) with its decomposition digits
split over the ranks, partial inner products summed by all_reduce over RCCL, the rest on every rank.
    python -m torch.distributed.run --nproc-per-node G tools/latency_mode.py [--config 4] [--count 1] [--reps 20]
One rank per GPU (nccl); with one GPU it runs at world size 1 (the collective still executes). Prints one JSON line on rank 0:
time per key switch unsplit on one GPU vs split over G, and whether the split result equals the unsplit one word for word."""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "gemini-seal_amd"))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402

import bench  # noqa: E402
import sealhip as S  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--config", type=int, default=4, choices=[3, 4, 5])
    ap.add_argument("--count", type=int, default=1, help="ciphertexts per key switch call (latency mode: 1)")
    ap.add_argument("--reps", type=int, default=20)
    a = ap.parse_args()
    rank, world = int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1"))
    local = int(os.environ.get("LOCAL_RANK", "0"))
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    os.environ.setdefault("MASTER_PORT", "29581")
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
    cfg = bench.CONFIGS[a.config]
    n, kmods = 1 << cfg["logn"], cfg["primes"]
    ctx = S.Context(cfg["scheme"], cfg["logn"], kmods, 1, cfg["t"], device=local)
    stream = torch.cuda.Stream(device=dev)
    ctx.set_stream(stream.cuda_stream)
    k, nk, m = ctx.k_first, len(kmods), a.count
    nd = ctx.kswitch_digits(k)
    with torch.cuda.stream(stream):
        torch.manual_seed(5)  # the same ciphertext and the same (replicated) key on every rank
        ct = torch.empty((m, 2, k, n), dtype=torch.int64, device=dev)
        target = torch.empty((m, k, n), dtype=torch.int64, device=dev)
        key = torch.empty((nd, 2, nk, n), dtype=torch.int64, device=dev)
        bench.fill_mod_rows(ct, kmods[:k])
        bench.fill_mod_rows(target, kmods[:k])
        bench.fill_mod_rows(key, kmods)
        dkey = S.KSwitchKeys(ctx, key, n_digits=nd, from_host=False)
        part = torch.empty((m, 2, k + ctx.nsp, n), dtype=torch.int64, device=dev)  # (k + nsp rows per component)
        unsplit = ct.clone()
        ctx.switch_key_inplace(k, unsplit, target, m, dkey)
    j0, j1 = bench.shard_range(nd, rank, world)

    def split_once(dst):
        ctx.switch_key_partial(k, target, m, dkey, j0, j1, part)
        stream.synchronize()  # the collective runs on its own stream
        dist.all_reduce(part, op=dist.ReduceOp.SUM)  # words below world * p < 2^63
        torch.cuda.current_stream().synchronize()
        ctx.switch_key_finish(k, dst, part, m, world)

    work = ct.clone()
    split_once(work)
    ctx.synchronize()
    same = bool(torch.equal(work, unsplit))

    def timed(fn):
        bench.barrier_sync()
        t0 = time.perf_counter()
        for _ in range(a.reps):
            fn()
        ctx.synchronize()
        bench.barrier_sync()
        return bench.max_over_ranks(time.perf_counter() - t0) / a.reps

    t_split = timed(lambda: split_once(work))
    t_one = timed(lambda: ctx.switch_key_inplace(k, work, target, m, dkey))
    ok = bench.all_ranks_true(same)
    if rank == 0:
        print(json.dumps({"mode": "latency (SURVEY 8e): digits of one key switch split over ranks", "config": a.config,
                          "ranks": world, "digits": nd, "digits_of_rank0": [j0, j1], "ciphertexts": m,
                          "ms_unsplit_one_gpu": t_one * 1e3, "ms_split": t_split * 1e3,
                          "all_reduce_bytes": part.numel() * 8, "split_equals_unsplit": ok}))
    dist.barrier()
    dist.destroy_process_group()
    if not ok:
        raise SystemExit("latency mode: split result differs from the unsplit key switch")


if __name__ == "__main__":
    main()
