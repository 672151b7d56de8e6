"""world_size-2 gloo rehearsal of bench.py's multi-GPU logic (no GPU needed): contiguous sharding of the
batch, barrier + max-over-ranks timing, digest gather to rank 0."""
import os
import sys

import torch
import torch.distributed as dist

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench  # noqa: E402


def main():
    dist.init_process_group("gloo")
    rank, world = dist.get_rank(), dist.get_world_size()
    total = 11
    lo, hi = bench.shard_range(total, rank, world)
    got = [None] * world
    dist.all_gather_object(got, (lo, hi))
    if rank == 0:
        assert got[0][0] == 0 and got[-1][1] == total
        for a, b in zip(got, got[1:]):
            assert a[1] == b[0]
    t = torch.tensor([1.0 + rank], dtype=torch.float64)
    tmax = bench.max_over_ranks(t.item())
    assert tmax == float(world)
    digests = bench.gather_digests(0x1000 + rank)
    assert bench.all_ranks_true(True) and not bench.all_ranks_true(rank == 0)
    # the final gather: every rank's slice lands on rank 0 and is checked there against the rank's own digest
    payload = torch.arange(rank * 1000, rank * 1000 + 96, dtype=torch.int64).reshape(3, 2, 4, 4)
    g = bench.gather_payload(payload, bench.cheap_digest)
    assert g["bytes_per_rank"] == 96 * 8 and g["backend"] == "gloo"
    if rank == 0:
        assert g["ranks_seen"] == world
        assert digests == [0x1000 + r for r in range(world)]
        print("SHARD_OK")
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
