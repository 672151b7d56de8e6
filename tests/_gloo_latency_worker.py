"""world_size-2 gloo rehearsal of SURVEY 8(e) "latency mode" (no GPU needed): the decomposition digits of ONE key switch are
split over the ranks, every rank forms the partial inner product of its digits (here: the CPU oracle's split key switch), the
partials are summed with all_reduce(SUM) on 64-bit words -- safe because ranks * p < 2^63 -- and the rest of the key switch runs
on the sum. The result must equal the unsplit key switch word for word."""
import ctypes as C
import os
import sys

import numpy as np
import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import bench  # noqa: E402
import oracle_lib as O  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    L = O.lib()
    ok = True
    for scheme, logn, bits, nsp in ((2, 10, [40, 40, 40, 41, 42], 1), (1, 10, [40, 40, 40, 40, 41], 2), (2, 11, [45] * 6, 3)):
        n = 1 << logn
        kmods = O.coeff_modulus_create(n, bits)
        ref = O.RefContext(scheme, logn, kmods, nsp=nsp, t=65537 if scheme == 1 else 0)
        k = ref.k_first
        nd = (k + nsp - 1) // nsp
        rng = np.random.default_rng(7)  # every rank draws the same ciphertext and key (the key is replicated)
        rows = lambda mods: np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in mods])  # noqa: E731
        key = np.stack([rows(kmods * 2).reshape(2, len(kmods), n) for _ in range(nd)])
        ct = rows(kmods[:k] * 2).reshape(2, k, n)
        target = rows(kmods[:k])
        j0, j1 = bench.shard_range(nd, rank, world)  # contiguous digit ranges, like the ciphertext shards
        part = np.zeros((2, k + nsp, n), dtype=np.uint64)
        assert L.ref_switch_key_partial(C.byref(ref.c), k, O.ptr(target), O.ptr(key), j0, j1, O.ptr(part)) == 0
        t = torch.from_numpy(part.view(np.int64).copy())
        dist.all_reduce(t, op=dist.ReduceOp.SUM)  # < world * p < 2^63: no wrap in int64
        summed = t.numpy().view(np.uint64)
        got = ct.copy()
        assert L.ref_switch_key_finish(C.byref(ref.c), k, O.ptr(got), O.ptr(summed)) == 0
        exp = ct.copy()
        assert L.ref_switch_key_inplace(C.byref(ref.c), k, O.ptr(exp), O.ptr(target), O.ptr(key)) == 0
        ok = ok and bool(np.array_equal(got, exp))  # (a rank may own no digit at all: its partial is zero)
    assert bench.all_ranks_true(ok)
    if rank == 0:
        print("LATENCY_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
