"""Raw PCIe rates of the box (pinned host memory, torch copies on two streams): H2D alone, D2H alone, both at once.
    python tools/pcie_probe.py [MiB per copy, default 512]
Context for bench.py's pcie_inclusive block: what the link gives when nothing else is in the way."""
import sys
import time

import torch

mib = int(sys.argv[1]) if len(sys.argv) > 1 else 512
n = mib << 20
dev = torch.device("cuda:0")
h_in = torch.empty(n, dtype=torch.uint8).pin_memory()
h_out = torch.empty(n, dtype=torch.uint8).pin_memory()
d_in = torch.empty(n, dtype=torch.uint8, device=dev)
d_out = torch.empty(n, dtype=torch.uint8, device=dev)
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()


def timed(fn, reps=8):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def h2d():
    with torch.cuda.stream(s1):
        d_in.copy_(h_in, non_blocking=True)


def d2h():
    with torch.cuda.stream(s2):
        h_out.copy_(d_out, non_blocking=True)


def both():
    h2d()
    d2h()


t = timed(h2d)
print("H2D alone   %.1f GB/s" % (n / t / 1e9))
t = timed(d2h)
print("D2H alone   %.1f GB/s" % (n / t / 1e9))
t = timed(both)
print("both at once: %.1f GB/s each way (%.1f total)" % (n / t / 1e9, 2 * n / t / 1e9))
for piece in (4, 1):
    m = piece << 20

    def pieces():
        with torch.cuda.stream(s1):
            for o in range(0, n, m):
                d_in[o:o + m].copy_(h_in[o:o + m], non_blocking=True)

    t = timed(pieces, 4)
    print("H2D in %d MiB pieces: %.1f GB/s" % (piece, n / t / 1e9))
