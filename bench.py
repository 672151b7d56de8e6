#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sealhip engine.

Metric (BASELINE.json): ciphertext multiply+relinearize/s (and forward-NTT/s) at N=2^15, 8 RNS primes.
A step = Evaluator::multiply + Evaluator::relinearize over one batch of independent synthetic BFV
ciphertexts that are already resident in HBM (config 3 of BASELINE.json: N=2^15, {55}x8 primes, k=7,
|Bsk|=8, 7 key digits, t=786433, PARITY mode = bit-exact with the reference).

    python bench.py --gpus N --steps K --warmup W [--batch B] [--config 3|4|5] [--mode parity|strict] [--force-dist]

--config selects the BASELINE.json line (default 3, the one the metric is quoted on; the driver's command is
unchanged): 4 = CKKS N=2^15, 12 primes, Evaluator::rotate_vector over the rank's share of the 8192 ciphertexts with the
replicated Galois key (evaluator.h:1201-1211); 5 = BFV N=2^16, 16 primes, multiply + relinearize + mod_switch_to_next
(evaluator.cpp:996-1036). Same JSON schema, same self-verification, same multi-rank logic for all three.

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL). When no launcher has started the ranks
(WORLD_SIZE unset) this process starts them itself as a child `python -m torch.distributed.run` -- before it
has touched the GPU -- and relays rank 0's JSON line. The batch of independent ciphertexts is sharded
contiguously over ranks with NO data-path collective (weak scaling: every rank owns `--batch` ciphertexts).
Collectives: the timing barrier / max-over-ranks, and after the timed region the final gather SURVEY 8(e) names
(every rank's output slice to rank 0 over RCCL/xGMI), timed separately and checked by digest.

--mode strict times SEALHIP_MODE_STRICT (SURVEY F4 "report both modes": Harvey-corrected forward butterflies and NTT'd
in-bundle BFV rows, the mode whose BFV results decrypt; evaluator.cpp:235-272 is what a user calls either way) with the
same schema, `config.mode` = "STRICT", verified against the oracle's STRICT restatement.

The timed path verifies itself: after the timed region at least 256 items of the batch (seeded pseudo-random picks plus
items 0, B/2, B-1 and the first and last item of every arena chunk the operations walked the batch in) are pulled
back to the host and compared word for word with the CPU oracle's result for the same inputs, on the thread pool the CPU
baseline uses (`verified_count`, `verified_items`); a mismatch fails the run.

`pcie_inclusive` (N = 1): the same step through the library's host-pointer entries on separately allocated pageable host
ciphertexts (what a std::vector<seal::Ciphertext> is), H2D and D2H included -- reported next to `value`, never as it.

Rank 0 prints one JSON line; see DESIGN.md "Measurement" for every field.
"""
import argparse
import ctypes as C
import hashlib
import json
import os
import socket
import subprocess
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
# CoeffModulus::Create(32768, {55}x8); re-derived by the oracle in the checker leg, hard-wired here so
# that the timed path never touches oracle/.
CFG3_PRIMES = [36028797010444289, 36028797012606977, 36028797013000193, 36028797013327873, 36028797014376449,
               36028797014573057, 36028797014704129, 36028797017456641]
# CoeffModulus::Create(32768, {50}x12) and (65536, {50}x16), checked against the oracle in the checker leg
CFG4_PRIMES = [1125899885412353, 1125899885740033, 1125899886395393, 1125899887312897, 1125899896160257, 1125899899174913,
               1125899901665281, 1125899902124033, 1125899903107073, 1125899903500289, 1125899903827969, 1125899904679937]
CFG5_PRIMES = [1125899864506369, 1125899865948161, 1125899870011393, 1125899870404609, 1125899877875713, 1125899879710721,
               1125899882987521, 1125899883380737, 1125899883642881, 1125899884036097, 1125899884167169, 1125899885740033,
               1125899886395393, 1125899887312897, 1125899902124033, 1125899903827969]

# The BASELINE.json lines bench.py can be timed on. `op`: what a step does to every ciphertext (pair) of the batch.
# Byte accounting per unit: `compulsory` = what any implementation has to move (inputs read once, result
# written once; keys are read once per batch and left out), `ntt_rows` = row transforms the reference performs for one
# unit (x 16 N bytes = "NTT-equivalent" traffic, so that fused kernels show up as a gain).
CONFIGS = {
    3: {"scheme": 1, "logn": 15, "bits": [55] * 8, "primes": CFG3_PRIMES, "t": PLAIN_T, "op": "mul_relin", "batch": 4096,
        "cpu_ops": 240, "unit": "ct_mul_relin/s",
        "metric": "ciphertext multiply+relinearize/s (BFV N=2^15, 8 primes, bit-exact PARITY mode)",
        "workload": "BASELINE config 3: BFV N=2^15, {55}x8 primes (k=7, |Bsk|=8, 7 digits), "
                    "Evaluator::multiply + relinearize over independent ciphertexts resident in HBM"},
    4: {"scheme": 2, "logn": 15, "bits": [50] * 12, "primes": CFG4_PRIMES, "t": 0, "op": "rotate", "batch": 1024,
        "cpu_ops": 160, "unit": "rotate_vector/s",
        "metric": "ciphertext rotate_vector/s (CKKS N=2^15, 12 primes, Galois key switch, bit-exact PARITY mode)",
        "workload": "BASELINE config 4: CKKS N=2^15, {50}x12 primes (k=11, 11 decomposition digits: 16 are not reachable with "
                    "12 primes, SURVEY 8), Evaluator::rotate_vector over the rank's share of 8192 ciphertexts resident in "
                    "HBM, Galois key replicated per GPU"},
    5: {"scheme": 1, "logn": 16, "bits": [50] * 16, "primes": CFG5_PRIMES, "t": PLAIN_T, "op": "mul_relin_modswitch",
        "batch": 256, "cpu_ops": 16, "unit": "pipeline/s",
        "metric": "multiply+relinearize+mod_switch_to_next pipelines/s (BFV N=2^16, 16 primes, bit-exact PARITY mode)",
        "workload": "BASELINE config 5: BFV N=2^16, {50}x16 primes (k=15, |Bsk|=16, 15 digits), Evaluator::multiply + "
                    "relinearize + mod_switch_to_next over independent ciphertexts resident in HBM"},
}


def unit_bytes(cfg):
    """Per-unit byte accounting of a config, in bytes."""
    n, nk = 1 << cfg["logn"], len(cfg["primes"])
    k, nsp = nk - 1, 1
    d = k  # ceil(k / nsp)
    if cfg["op"] == "rotate":
        compulsory = 2 * (2 * k * n * 8)  # read the ciphertext, write the rotated one
        ntt_rows = k + d * (k + nsp - 1) + 2 * nsp + 2 * k  # target iNTT + digit NTTs + special iNTTs + temp NTTs (CKKS)
    else:
        compulsory = 48 * k * n  # two size-2 inputs read, one size-2 result written
        ntt_rows = 7 * (2 * k + 1) + d * (k + nsp - 1) + 2 * nsp + 2 * k  # BFV multiply + relinearize (cfg3: 105 + 65)
        if cfg["op"] == "mul_relin_modswitch":
            compulsory = 2 * (2 * k * n * 8) + 2 * (k - 1) * n * 8
    return {"compulsory": compulsory, "ntt_rows": ntt_rows, "ntt_equivalent": ntt_rows * 16 * n}


def kernel_bytes_per_unit(cfg, tag):
    """Own algorithmic bytes of the non-NTT kernels per unit (rows read once + rows written once, 8 N bytes per row;
    DESIGN.md section 4), or None. The NTT tags are priced from the rows their launches report."""
    n, nk = 1 << cfg["logn"], len(cfg["primes"])
    k, nb, rows, d = nk - 1, nk, nk, nk - 1  # |Bsk| = k + 1 = nk, key rows = k + nsp
    row = 8 * n
    table = {
        "bfv_lift": 4 * (k + nb) * row,                       # 4 polynomials: k rows in, |Bsk| rows out
        "bfv_floor_sk": 3 * (k + nb + k) * row,               # 3 polynomials: k + |Bsk| rows in, k rows out
        "ks_mac": (d * rows + 2 * rows) * row,                # digit rows in, both product polynomials out (key: per batch)
        "ks_moddown_bfv": (2 * rows + 2 * k + 2 * k) * row,   # products in, ciphertext in and out
        "ks_moddown_post": (2 * k + 2 * k + 2 * k + 2 * k) * row,
        "galois": (2 * k + 2 * k) * row,
        "divround_bfv": (2 * k + 2 * (k - 1)) * row,
        "tensor_product": 7 * k * row,
    }
    return table.get(tag)


# ------------------------------------------------------------------ multi-rank helpers (covered by gloo tests)
def shard_range(total, rank, world):
    """Contiguous shard [lo, hi) of `total` independent ciphertexts for `rank`."""
    return (rank * total) // world, ((rank + 1) * total) // world


def _dist_on():
    return dist.is_available() and dist.is_initialized()


def _coll_device():
    return "cuda" if dist.get_backend() == "nccl" else "cpu"


def max_over_ranks(seconds):
    if not _dist_on():
        return seconds
    t = torch.tensor([seconds], dtype=torch.float64, device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    return float(t.item())


def all_ranks_true(flag):
    if not _dist_on():
        return bool(flag)
    t = torch.tensor([1 if flag else 0], dtype=torch.int64, device=_coll_device())
    dist.all_reduce(t, op=dist.ReduceOp.MIN)
    return bool(t.item())


def gather_digests(digest):
    """8-byte digests of every rank to every rank (cross-check of the payload gather)."""
    if not _dist_on():
        return [digest]
    mine = torch.tensor([digest], dtype=torch.int64, device=_coll_device())
    out = [torch.zeros_like(mine) for _ in range(dist.get_world_size())]
    dist.all_gather(out, mine)
    return [int(t.item()) for t in out]


def barrier_sync():
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()
    if _dist_on():
        dist.barrier()
    if torch.cuda.is_available() and torch.cuda.is_initialized():
        torch.cuda.synchronize()


def gather_payload(payload, digest_fn, force=False):
    """The final gather of SURVEY 8(e): every rank's output slice to rank 0 (RCCL send/recv over xGMI under the nccl
    backend), timed on its own, then checked on rank 0 against the digests the ranks computed locally.
    Returns None for a single rank -- unless `force` (--force-dist): then the one rank gathers to itself, so that the
    collective really runs on the backend (the rehearsal of the multi-GPU path on a one-GPU box)."""
    if not _dist_on() or (dist.get_world_size() == 1 and not force):
        return None
    world, rank = dist.get_world_size(), dist.get_rank()
    mine = digest_fn(payload)
    digests = gather_digests(mine)
    bufs = [torch.empty_like(payload) for _ in range(world)] if rank == 0 else None
    barrier_sync()
    t0 = time.perf_counter()
    dist.gather(payload, gather_list=bufs, dst=0)
    barrier_sync()
    dt = max_over_ranks(time.perf_counter() - t0)
    nbytes = payload.numel() * payload.element_size()
    seen = 0
    if rank == 0:
        seen = sum(1 for r in range(world) if digest_fn(bufs[r]) == digests[r])
    return {"bytes_per_rank": nbytes, "seconds": dt, "GBps_into_root": max(world - 1, 1) * nbytes / dt / 1e9,
            "ranks_seen": seen, "backend": dist.get_backend()}


def cheap_digest(t):
    """Order-dependent 63-bit digest computed where the tensor lives."""
    flat = t.reshape(-1)
    idx = torch.arange(flat.numel(), device=flat.device, dtype=torch.int64)
    return int(((flat * 0x9E3779B97F4A7C15 + idx) ^ (flat >> 29)).sum().item() & 0x7FFFFFFFFFFFFFFF)


def spawn_ranks(args):
    """`python bench.py --gpus N` without a launcher: start the N ranks as a child process (this process has not
    initialised the GPU and never execs) and relay what they print."""
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=%d" % args.gpus,
           "--master-addr", "127.0.0.1", "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
    env.setdefault("OMP_NUM_THREADS", "1")
    proc = subprocess.Popen(cmd, stdout=subprocess.PIPE, env=env, text=True)
    for line in proc.stdout:
        sys.stdout.write(line)
        sys.stdout.flush()
    return proc.wait()


# ------------------------------------------------------------------ synthetic data
def fill_mod_rows(t, moduli):
    """t[..., i, :] uniform in [0, moduli[i]) -- the distribution real ciphertexts/keys have."""
    for i, p in enumerate(moduli):
        view = t[..., i, :]
        view.copy_(torch.randint(0, p, view.shape, dtype=torch.int64, device=t.device))


def host_u64(t):
    return np.ascontiguousarray(t.detach().cpu().numpy()).view(np.uint64)


# ------------------------------------------------------------------ the checker leg (oracle = CPU "port")
def cpu_info():
    model, phys = "unknown", set()
    try:
        pid = core = None
        for ln in open("/proc/cpuinfo"):
            if ln.startswith("model name") and model == "unknown":
                model = ln.split(":", 1)[1].strip()
            elif ln.startswith("physical id"):
                pid = ln.split(":", 1)[1].strip()
            elif ln.startswith("core id"):
                core = ln.split(":", 1)[1].strip()
                phys.add((pid, core))
    except OSError:
        pass
    quota = None
    try:  # cgroup v2 CPU quota of this container, in cores
        a, b = open("/sys/fs/cgroup/cpu.max").read().split()
        quota = None if a == "max" else float(a) / float(b)
    except (OSError, ValueError):
        pass
    try:
        affinity = len(os.sched_getaffinity(0))
    except (AttributeError, OSError):
        affinity = None
    return {"cpu_model": model, "logical_cpus": os.cpu_count(), "physical_cores": len(phys) or None,
            "affinity_cpus": affinity, "cgroup_cpu_quota": quota}


def load_oracle():
    """The CPU oracle, rebuilt for THIS host with -O3 -march=native when a compiler is present; the
    portable build otherwise. Returns (module, how it was built)."""
    how = "gcc -O3 (portable build shipped with the repo)"
    try:
        flags = [ln for ln in open("/proc/cpuinfo") if ln.startswith("flags")][:1]
        tag = hashlib.sha1((cpu_info()["cpu_model"] + "".join(flags)).encode()).hexdigest()[:12]
        d = os.path.join(ROOT, "oracle", "_native")
        os.makedirs(d, exist_ok=True)
        path = os.path.join(d, "libsealref_native_%s.so" % tag)
        src = os.path.join(ROOT, "oracle", "sealref.c")
        if not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
            tmp = path + ".%d.tmp" % os.getpid()
            subprocess.check_call(["gcc", "-O3", "-march=native", "-fPIC", "-std=gnu11", "-ffp-contract=off", "-shared",
                                   "-o", tmp, src, "-lm"], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
            os.replace(tmp, path)
        os.environ["SEALREF_LIBRARY"] = path
        how = "gcc -O3 -march=native, built on this host"
    except Exception:  # no compiler / read-only tree: the portable build
        os.environ.pop("SEALREF_LIBRARY", None)
    import oracle_lib as O

    O.lib()
    return O, how


class OracleOp:
    """One unit of a config's step on the CPU oracle (oracle/sealref.c): the checker of the timed path's output and the
    thing the CPU baseline times. `run(a, b, reps)` -> the result words after applying the step `reps` times (the in-place
    rotate of config 4 is applied once per step, so the timed output is the (warmup + steps)-fold rotation)."""

    def __init__(self, O, cfg, key_host, galois_elt=None, strict=False):
        self.O, self.L, self.cfg = O, O.lib(), cfg
        self.n = 1 << cfg["logn"]
        kmods = O.coeff_modulus_create(self.n, cfg["bits"])
        assert kmods == cfg["primes"], "hard-wired primes differ from CoeffModulus::Create"
        self.ref = O.RefContext(cfg["scheme"], cfg["logn"], kmods, nsp=NSP, t=cfg["t"], mode=1 if strict else 0)
        self.k = self.ref.k_first
        self.key = key_host
        self.keys = (C.c_void_p * 1)(key_host.ctypes.data)
        self.elt = galois_elt
        if cfg["scheme"] == 1:
            self.ref.rns_tool(self.k)  # shared constants are built before any worker thread starts

    def run(self, a, b, reps=1):
        O, L, ref, k, n = self.O, self.L, self.ref, self.k, self.n
        op = self.cfg["op"]
        if op == "rotate":
            x = a.copy()
            for _ in range(reps):
                assert L.ref_apply_galois_inplace(C.byref(ref.c), k, O.ptr(x), self.elt, O.ptr(self.key)) == 0
            return x
        out = np.zeros((3, k, n), dtype=np.uint64)
        assert L.ref_bfv_multiply(C.byref(ref.c), k, O.ptr(a), 2, O.ptr(b), 2, O.ptr(out)) == 0
        assert L.ref_relinearize(C.byref(ref.c), k, O.ptr(out), 3, self.keys) == 0
        if op == "mul_relin":
            return out
        c2 = np.ascontiguousarray(out[:2])
        ms = np.zeros((2, k - 1, n), dtype=np.uint64)
        assert L.ref_mod_switch_scale_to_next(C.byref(ref.c), k, O.ptr(c2), 2, O.ptr(ms)) == 0
        return ms


def usable_cpus():
    """CPUs this process may really use: min(logical CPUs, scheduler affinity, cgroup CPU quota)."""
    import math

    info = cpu_info()
    usable = [c for c in (info["logical_cpus"], info["affinity_cpus"],
                          math.ceil(info["cgroup_cpu_quota"]) if info["cgroup_cpu_quota"] else None) if c]
    return max(1, min(usable) if usable else 1)


def cpu_baseline(O, how, cfg, total_ops, galois_elt, strict=False):
    """Times the CPU oracle (oracle/sealref.c, digest-identical to the reference) on a bounded sample of the SAME
    workload: one worker thread per CPU this process may use -- min(logical CPUs, scheduler affinity, cgroup CPU quota);
    threads beyond the quota only get throttled -- each running the config's step on its own ciphertexts (the
    reference is single-threaded per call and thread-safe across calls). The host's CPU model and core counts are
    reported next to it, with the linear projection of the 1-thread rate to every physical core of the host (an upper
    bound: it assumes perfect scaling), because a GPU box hands its container only a share of the host's cores."""
    import math
    from concurrent.futures import ThreadPoolExecutor

    L = O.lib()
    info = cpu_info()
    threads = usable_cpus()
    per_thread = max(1, int(round(total_ops / threads)))
    n, kmods = 1 << cfg["logn"], cfg["primes"]
    k = len(kmods) - NSP
    rng = np.random.default_rng(1)

    def rows(mods):
        return np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in mods])

    key = np.stack([rows(kmods * 2).reshape(2, len(kmods), n) for _ in range(k)])
    op = OracleOp(O, cfg, key, galois_elt, strict)
    work = [(rows(kmods[:k] * 2).reshape(2, k, n), rows(kmods[:k] * 2).reshape(2, k, n)) for _ in range(threads)]

    def run(item, reps=per_thread):
        for _ in range(reps):
            op.run(item[0], item[1])

    run(work[0], 1)  # page everything in
    t1 = time.perf_counter()
    one_reps = 4 if cfg["logn"] < 16 else 1
    run(work[0], one_reps)
    one_thread = one_reps / (time.perf_counter() - t1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, work))
    dt = time.perf_counter() - t0
    # forward NTT/s on one core, same primes
    x = rows(kmods[:k])
    t1 = time.perf_counter()
    reps = 8 if cfg["logn"] < 16 else 2
    for _ in range(reps):
        for i in range(k):
            L.ref_ntt_forward(O.ptr(x[i]), op.ref.tables(i), 0)
    ntt_s = reps * k / (time.perf_counter() - t1)
    out = {
        "value": threads * per_thread / dt,
        "unit": cfg["unit"],
        "cores": threads,
        "kind": "port",
        "sample": "%d threads (one per CPU usable by this container) x %d x the step of this config on the CPU oracle "
                  "(same workload, %d units)" % (threads, per_thread, threads * per_thread),
        "seconds": dt,
        "value_1thread": one_thread,
        "forward_ntt_per_s_1core": ntt_s,
        "build": how,
    }
    out.update(info)
    cores = info["physical_cores"] or info["logical_cpus"] or threads
    out["projected_all_physical_cores_linear"] = one_thread * cores
    out["projection_note"] = ("the >= 10x CPU target is judged against a linear PROJECTION of the 1-thread rate to all %s "
                              "physical cores (the container may use %d CPUs); it assumes perfect scaling of a "
                              "memory-heavy workload, i.e. it is an upper bound on the host" % (cores, threads))
    return out


# ------------------------------------------------------------------ workloads
class EngineWorkload:
    """One BASELINE config on one MI355X through the C ABI (ctypes mirror of the Evaluator interface)."""

    def __init__(self, args, rank, local_rank):
        import sealhip as S

        if not torch.cuda.is_available() or S.num_devices() < 1:
            raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.cfg = cfg = CONFIGS[args.config]
        self.dev = torch.device("cuda", local_rank)
        # one explicit stream for torch's fills/copies AND the engine's launches: everything below is ordered on it
        self.stream = torch.cuda.Stream(device=self.dev)
        self.n, self.kmods = 1 << cfg["logn"], cfg["primes"]
        self.strict = args.mode == "strict"
        self.ctx = S.Context(cfg["scheme"], cfg["logn"], self.kmods, NSP, cfg["t"],
                             mode=S.MODE_STRICT if self.strict else S.MODE_PARITY, device=local_rank)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.ev = S.Evaluator(self.ctx)
        self.S = S
        self.k, self.nk, self.B = self.ctx.k_first, len(self.kmods), args.batch
        self.steps_done = 0
        self.elt = self.ctx.galois_elt_from_step(1) if cfg["op"] == "rotate" else None
        k, n, B, dev = self.k, self.n, self.B, self.dev
        with torch.cuda.stream(self.stream):
            torch.manual_seed(1234 + rank)  # every rank owns different ciphertexts
            self.a = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
            fill_mod_rows(self.a, self.kmods[:k])
            self.key = torch.empty((k, 2, self.nk, n), dtype=torch.int64, device=dev)
            if cfg["op"] == "rotate":
                self.b = self.out = None
            else:
                self.b = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
                self.out = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
                fill_mod_rows(self.b, self.kmods[:k])
            self.out2 = None
            if cfg["op"] == "mul_relin_modswitch":
                self.out2 = torch.empty((B, 2, k - 1, n), dtype=torch.int64, device=dev)
            torch.manual_seed(99)  # the relinearisation / Galois key is replicated on every GPU
            fill_mod_rows(self.key, self.kmods)
            self.rk = S.KSwitchKeys(self.ctx, self.key, n_digits=k, from_host=False)
        self.want_items = 0 if args.no_verify else max(3, args.verify_items)
        self.items, self.saved_a, self.chunks, self.saved_at = [], {}, [], 0

    def choose_items(self):
        """Which items the checker leg verifies: 0, B/2, B-1, the first and last item of every arena chunk the operations of
        a step walk the batch in (the launch geometry is position dependent: chunked launches, XCD-aware block maps, item
        groups), and seeded pseudo-random picks up to --verify-items. Called after the first warm-up step (the chunk
        plan is what the library reports, sealhip_debug_chunk_log) and before the timed region; config 4 rotates in place,
        so the inputs of the chosen items are kept."""
        B = self.B
        picks = {0, B // 2, B - 1}
        try:
            self.chunks = sorted({c for cnt, c in self.ctx.chunk_log() if cnt == B and 0 < c < B})
        except Exception:  # (a library without the debug entry: the edges are simply not among the picks)
            self.chunks = []
        for c in self.chunks:
            for edge in range(c, B, c):
                picks.update((edge - 1, edge))
        rng = np.random.default_rng(20261005 + B)
        for i in rng.permutation(B):
            if len(picks) >= min(B, self.want_items):
                break
            picks.add(int(i))
        self.items = sorted(picks)
        self.saved_at = self.steps_done
        if self.cfg["op"] == "rotate":
            with torch.cuda.stream(self.stream):
                self.saved_a = {i: host_u64(self.a[i]) for i in self.items}

    def step(self):
        op, k, B = self.cfg["op"], self.k, self.B
        if op == "rotate":
            self.ev.rotate_vector_inplace(self.a, k, B, 1, {self.elt: self.rk})
        else:
            self.ev.multiply(self.a, 2, self.b, 2, k, B, self.out)
            self.ev.relinearize_inplace(self.out, 3, k, B, [self.rk])
            if op == "mul_relin_modswitch":
                # the relinearized ciphertext is the first two polynomials of each size-3 item: the strided entry reads it
                # where it lies (no compaction pass)
                self.ev.mod_switch_to_next(self.out, 2, k, B, self.out2, item_stride=3 * k * self.n)
        self.steps_done += 1

    def finish(self):
        self.ctx.synchronize()  # also surfaces a device-side failure of any launch (sticky flag)

    def results(self):
        op = self.cfg["op"]
        return self.a if op == "rotate" else (self.out2 if op == "mul_relin_modswitch" else self.out)

    def result_slice(self, count):
        with torch.cuda.stream(self.stream):
            r = self.results()
            res = r[:count, :2].contiguous()  # the size-2 results, compacted (what a caller gets back)
        self.stream.synchronize()  # the callers (digests, RCCL gather) work on other streams
        return res

    def key_digest(self):
        with torch.cuda.stream(self.stream):
            return cheap_digest(self.key[0, 0, :1])

    def verify(self, O):
        """The timed path's own output for the chosen items against the CPU oracle, word for word, on one worker thread
        per usable CPU (the oracle is thread-safe across calls, like the reference). -> list of items that differ."""
        from concurrent.futures import ThreadPoolExecutor

        with torch.cuda.stream(self.stream):
            op = OracleOp(O, self.cfg, host_u64(self.key), self.elt, self.strict)
            rotate = self.cfg["op"] == "rotate"
            res = self.results()

            def fetch(i):
                got = host_u64(res[i])
                a = self.saved_a[i] if rotate else host_u64(self.a[i])
                b = None if rotate else host_u64(self.b[i])
                return i, got, a, b

            def check(job):
                i, got, a, b = job
                exp = op.run(a, b, reps=self.steps_done - self.saved_at) if rotate else op.run(a, b)
                return i, bool(np.array_equal(got[: exp.shape[0]], exp))

            bad = []
            with ThreadPoolExecutor(max_workers=usable_cpus()) as ex:
                # (fetched in slices so that a few hundred 5 MB items never sit in host memory all at once)
                for lo in range(0, len(self.items), 64):
                    jobs = [fetch(i) for i in self.items[lo:lo + 64]]
                    bad += [i for i, ok in ex.map(check, jobs) if not ok]
        return bad

    def pcie_inclusive(self, pairs):
        """SURVEY 8(d) "report separately with H2D/D2H included": the same step through the library's host-pointer entries
        (csrc/hostbatch.cpp: gather threads, pinned double-buffered staging, H2D / compute / D2H on three streams) on
        `pairs` SEPARATELY ALLOCATED pageable host ciphertexts -- what a std::vector<seal::Ciphertext> is
        (ciphertext.h:709-721). Results are checked against the device-resident path on the same inputs."""
        op, k, n, ev = self.cfg["op"], self.k, self.n, self.ev
        P = min(pairs, self.B)
        if P <= 0:
            return None
        with torch.cuda.stream(self.stream):
            ha = [host_u64(self.a[i]).copy() for i in range(P)]
            hb = None if op == "rotate" else [host_u64(self.b[i]).copy() for i in range(P)]
            if op == "rotate":  # the device path on a copy of the same inputs
                ref = self.a[:P].clone()
                ev.rotate_vector_inplace(ref, k, P, 1, {self.elt: self.rk})
                ref = host_u64(ref)
            else:
                ref = host_u64(self.results()[:P])
        self.ctx.synchronize()
        kk = k - 1 if op == "mul_relin_modswitch" else k
        ho = None if op == "rotate" else [np.zeros((2, k, n), dtype=np.uint64) for _ in range(P)]
        ho2 = [np.zeros((2, kk, n), dtype=np.uint64) for _ in range(P)] if op == "mul_relin_modswitch" else None

        def run(work, ha=ha, hb=hb, ho=ho, ho2=ho2):
            if op == "rotate":
                ev.rotate_vector_host(work, k, 1, {self.elt: self.rk})
            else:
                ev.multiply_host(ha, 2, hb, 2, k, ho, relin_keys=[self.rk])
                if op == "mul_relin_modswitch":
                    ev.mod_switch_to_next_host(ho, 2, k, ho2)

        first = [x.copy() for x in ha] if op == "rotate" else None
        run(first)  # staging buffers, arena; also the run that is compared
        got = first if op == "rotate" else (ho2 if ho2 else ho)
        ok = all(np.array_equal(got[i], ref[i][:2] if op != "rotate" else ref[i]) for i in range(P))
        reps = 2
        works = [[x.copy() for x in ha] if op == "rotate" else None for _ in range(reps)]  # (the in-place op's inputs: not timed)
        t0 = time.perf_counter()
        for wk in works:
            run(wk)
        dt = (time.perf_counter() - t0) / reps
        in_item = 2 * k * n * 8 * (1 if op == "rotate" else 2)
        out_item = 2 * kk * n * 8
        res = {"value": P / dt, "unit": self.cfg["unit"], "units": P, "seconds_per_call": dt,
               "boundary": "sealhip_evaluator_*_host: %d separately allocated pageable host ciphertext%s in, results out "
                           "(gather threads + pinned double-buffered staging + H2D / compute / D2H streams inside the "
                           "library)" % (P, "s" if op == "rotate" else " pairs"),
               "h2d_GBps": P * in_item / dt / 1e9, "d2h_GBps": P * out_item / dt / 1e9, "matches_device_path": bool(ok),
               "note": "PCIe-inclusive rate, reported next to `value` (inputs resident in HBM), never as it"}
        del works, first
        # The same entries on ciphertexts that are pieces of REGISTERED pool blocks (sealhip_host_register: what an integration
        # that pins the MemoryPool's allocations once gets, mempool.cpp:45,145): no staging copy, the DMA engine reads / writes
        # the caller's buffers. Pinning is done once and not timed (its cost is reported); at most ~3 GB are pinned here.
        try:
            per_item = in_item + 2 * k * n * 8 + (out_item if op == "mul_relin_modswitch" else 0)
            Q = max(1, min(P, int(3e9 // (per_item * (reps + 1 if op == "rotate" else 1)))))
            blocks = []

            def pooled(src, shape):
                items = []
                for lo in range(0, Q, 32):
                    blk = np.zeros((min(32, Q - lo),) + shape, dtype=np.uint64)
                    blocks.append(blk)
                    for j in range(blk.shape[0]):
                        if src is not None:
                            blk[j][...] = src[lo + j]
                        items.append(blk[j])
                return items

            if op == "rotate":
                sets = [pooled(ha, (2, k, n)) for _ in range(reps + 1)]
                ra = rb = ro = ro2 = None
            else:
                ra, rb = pooled(ha, (2, k, n)), pooled(hb, (2, k, n))
                ro = pooled(None, (2, k, n))
                ro2 = pooled(None, (2, kk, n)) if op == "mul_relin_modswitch" else None
                sets = [None] * (reps + 1)
            t0 = time.perf_counter()
            for blk in blocks:
                ev.host_register(blk)
            t_reg = time.perf_counter() - t0
            try:
                run(sets[0], ra, rb, ro, ro2)
                got = sets[0] if op == "rotate" else (ro2 if ro2 else ro)
                rok = all(np.array_equal(got[i], ref[i][:2] if op != "rotate" else ref[i]) for i in range(Q))
                t0 = time.perf_counter()
                for wk in sets[1:]:
                    run(wk, ra, rb, ro, ro2)
                rdt = (time.perf_counter() - t0) / reps
            finally:
                for blk in blocks:
                    ev.host_unregister(blk)
            res["registered"] = {"value": Q / rdt, "units": Q, "seconds_per_call": rdt, "h2d_GBps": Q * in_item / rdt / 1e9,
                                 "d2h_GBps": Q * out_item / rdt / 1e9, "matches_device_path": bool(rok),
                                 "pinned_bytes": int(sum(b.nbytes for b in blocks)), "register_seconds": t_reg,
                                 "boundary": "the same entries on ciphertexts that are pieces of pool blocks pinned in place once "
                                             "(sealhip_host_register; not timed): no staging copies"}
        except Exception as exc:  # an auxiliary figure must not cost the run its line
            res["registered"] = {"error": "%s: %s" % (type(exc).__name__, exc)}
        return res


class StubWorkload:
    """CPU stand-in with the same rank logic (seeds, replicated key, step, result slice) for the gloo tests of the
    launcher: NOT the engine and never timed as such (`"stub": true` in the line). --config changes the shape of the
    stand-in's step the way it changes the engine's (in-place rotate for 4, a third stage for 5)."""

    def __init__(self, args, rank, local_rank):
        self.cfg = CONFIGS[args.config]
        self.dev = torch.device("cpu")
        self.n, self.kmods = 64, [1073479681, 1073184769, 1072496641]
        self.k, self.nk, self.B = 2, 3, args.batch
        g = torch.Generator().manual_seed(1234 + rank)
        self.a = torch.randint(0, self.kmods[0], (self.B, 2, self.k, self.n), dtype=torch.int64, generator=g)
        self.b = torch.randint(0, self.kmods[0], (self.B, 2, self.k, self.n), dtype=torch.int64, generator=g)
        self.out = torch.zeros((self.B, 3, self.k, self.n), dtype=torch.int64)
        g = torch.Generator().manual_seed(99)
        self.key = torch.randint(0, self.kmods[0], (self.k, 2, self.nk, self.n), dtype=torch.int64, generator=g)
        self.local_rank = local_rank

    def step(self):
        op = self.cfg["op"]
        if op == "rotate":
            self.a = (torch.roll(self.a, 1, dims=-1) + self.key[0, 0, 0, 0]) % self.kmods[0]
            return
        self.out[:, :2] = (self.a * self.b + self.key[0, 0, 0, 0]) % self.kmods[0]
        if op == "mul_relin_modswitch":
            self.out[:, :2] = (self.out[:, :2] * 3) % self.kmods[0]

    def finish(self):
        pass

    def result_slice(self, count):
        src = self.a if self.cfg["op"] == "rotate" else self.out
        return src[:count, :2].contiguous()

    def key_digest(self):
        return cheap_digest(self.key[0, 0, :1])


# ------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--config", type=int, default=3, choices=sorted(CONFIGS),
                    help="BASELINE.json line: 3 = BFV multiply+relinearize (the metric's config, default), 4 = CKKS "
                         "rotate_vector, 5 = BFV N=2^16 multiply+relinearize+mod_switch_to_next")
    ap.add_argument("--mode", choices=("parity", "strict"), default="parity",
                    help="parity (default): bit-exact with the reference as built, SURVEY F2/F3 included; strict: "
                         "SEALHIP_MODE_STRICT, the mode whose BFV results decrypt (SURVEY B.6) -- same schema")
    ap.add_argument("--verify-items", type=int, default=256,
                    help="items of the batch checked word for word against the CPU oracle after the timed region")
    ap.add_argument("--pcie-pairs", type=int, default=256,
                    help="separately allocated pageable host ciphertexts in the PCIe-inclusive section (0: skip)")
    ap.add_argument("--batch", type=int, default=None,
                    help="independent ciphertexts (pairs) per GPU; default per config: 4096 / 1024 / 256 "
                         "(SEALHIP_BENCH_BATCH overrides the default)")
    ap.add_argument("--ntt-polys", type=int, default=8192,
                    help="polynomials (x k rows) in one launch of the NTT-only section (a quarter of it at N = 2^16). The "
                         "rate grows with the launch until ~57 k rows (round 3, same box: 35.1 %% of the roofline at 14 k "
                         "rows, 38.2 %% at 29 k, 39.5 %% at 57 k and at 115 k): 8192 x 7 rows is the steady state")
    ap.add_argument("--gather-cts", type=int, default=512,
                    help="size-2 result ciphertexts per rank in the final gather to rank 0 (N > 1 only; 3.67 MB each)")
    ap.add_argument("--cpu-ops", type=int, default=None,
                    help="units of the CPU baseline sample (default per config: 240 / 160 / 16, about 10-30 s of CPU work)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed path's output")
    ap.add_argument("--force-dist", action="store_true",
                    help="initialise torch.distributed (nccl = RCCL) and run every collective of the multi-rank path even "
                         "with one rank: barrier, all_reduce, all_gather and the final gather (to self)")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="first collect the HBM PMC counters of the step (two rocprofv3 --pmc child runs of this "
                         "command at batch 256) and record them in profiles/traffic.json; roofline.traffic then comes "
                         "from this run's own measurement")
    ap.add_argument("--stub", action="store_true", help="CPU stand-in workload + gloo (tests of the rank logic only)")
    args = ap.parse_args(argv)
    cfg = CONFIGS[args.config]
    if args.batch is None:
        env = os.environ.get("SEALHIP_BENCH_BATCH")
        args.batch = int(env) if env and args.config == 3 else cfg["batch"]
    if args.cpu_ops is None:
        args.cpu_ops = cfg["cpu_ops"]
    return args


def main(argv=None):
    args = parse_args(argv)
    cfg = CONFIGS[args.config]
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if not args.stub and args.gpus > torch.cuda.device_count():  # (counting devices does not initialise the GPU)
        raise SystemExit("bench.py: --gpus %d but this node shows %d HIP device(s): nothing was launched"
                         % (args.gpus, torch.cuda.device_count()))
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # nothing above has touched the GPU
    if args.measure_traffic and args.gpus == 1 and not args.stub:
        measure_traffic(args)  # child processes; this one has not touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match the %d launched ranks (WORLD_SIZE)" % (args.gpus, world))
    if world > 1 or args.force_dist:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if "MASTER_PORT" not in os.environ:  # --force-dist without a launcher: a one-rank group of our own
            with socket.socket() as sk:
                sk.bind(("127.0.0.1", 0))
                os.environ["MASTER_PORT"] = str(sk.getsockname()[1])
        os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")  # dmabuf IPC: RCCL needs it on this pool
        if args.stub:
            dist.init_process_group("gloo", rank=rank, world_size=world)
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", local_rank))
    w = (StubWorkload if args.stub else EngineWorkload)(args, rank, local_rank)
    B, k, n = w.B, w.k, w.n
    lo, hi = shard_range(world * B, rank, world)  # this rank's slice of the global batch (weak scaling: B each)
    assert hi - lo == B

    for _ in range(args.warmup):
        w.step()
        if not args.stub and not w.items:
            w.choose_items()  # after the first step: the arena chunks are known (and the in-place config keeps its inputs)
    if not args.stub and not w.items:
        w.choose_items()
    barrier_sync()
    if not args.stub:
        w.ctx.profile_enable(True)
    t0 = time.perf_counter()
    for _ in range(args.steps):
        w.step()
    barrier_sync()
    dt = time.perf_counter() - t0
    prof = {}
    if not args.stub:
        prof = w.ctx.profile_fetch()
        w.ctx.profile_enable(False)
    w.finish()
    dt = max_over_ranks(dt)
    value = world * B * args.steps / dt

    # ---- roofline of the dominant kernel, from HIP events recorded on the launch stream during the timed steps
    total_ms = sum(v["ms"] for v in prof.values()) or 1.0
    shares = {tkey: round(v["ms"] / total_ms, 4) for tkey, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])}

    # launches per full row transform: the single-pass kernels complete a transform per launch, the tiled
    # pass kernel needs two launches (each then counts for half of the 16*N algorithmic bytes)
    PASSES = {"ntt_fwd_half": 1, "ntt_inv_half": 1, "ntt_fwd_pass": 2, "ntt_inv_pass": 2}

    def ntt_roofline(tag):
        v = prof.get(tag)
        if not v or not v["launches"]:
            return None
        rows_per_launch = v["units"] / v["launches"]
        avg_s = v["ms"] / v["launches"] / 1e3
        alg_bytes = rows_per_launch * 16 * n / PASSES[tag]  # 16*N bytes per row transform
        ach = alg_bytes / avg_s / 1e9
        return {"kernel": tag, "bound": "hbm", "achieved": ach, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                "frac": ach / HBM_PEAK_GBS, "traffic": None, "avg_launch_ms": avg_s * 1e3,
                "rows_per_launch": rows_per_launch, "algorithmic_bytes_per_launch": alg_bytes,
                "launches": v["launches"], "share_of_step_kernel_time": v["ms"] / total_ms}

    def valu_ceiling(r):
        """SURVEY 8(d) "secondary limiter": what the transform's own arithmetic allows on THIS device -- the rate of a
        kernel that executes nothing but the butterfly sequence of the launches' instance (sealhip_debug_butterfly_rate:
        same instructions, operands in registers, same occupancy), divided by the N/2 log2 N butterflies of a row. The
        integer instances are bound by it (vector-ALU issue under the package power cap), not by HBM."""
        if args.stub or not r or not r.get("achieved"):
            return
        try:
            fp = all(p < (1 << 50) for p in w.kmods[:k]) and cfg["scheme"] == 2
            kind = 3 if fp else (0 if args.mode == "strict" else 2)
            rate = w.ctx.butterfly_rate(kind, 0)
            rate_exact = rate if fp else w.ctx.butterfly_rate(0, 0)
        except Exception:
            return
        per_row = (n // 2) * cfg["logn"]
        rows_s = rate / per_row
        r["bound"] = "hbm" if fp else "valu"
        r["valu_ceiling"] = {
            "butterfly": {3: "FP64 (primes below 2^50)", 2: "integer, approximate Shoup quotient level 2", 0: "integer, exact Shoup quotient"}[kind],
            "butterflies_per_s": rate, "butterflies_per_s_reference_sequence": rate_exact,
            "butterflies_per_row": per_row, "rows_per_s": rows_s, "as_frac_of_hbm": rows_s * 16 * n / 1e9 / HBM_PEAK_GBS,
            "note": "measured in this run on this device: a kernel of nothing but the instance's butterflies, operands in registers"}
        r["alu_ceiling_frac"] = r["frac"] / r["valu_ceiling"]["as_frac_of_hbm"]
        # With the counters of --measure-traffic (SQ_INSTS_VALU in its own --pmc pass): the kernel executes more vector
        # instructions than its butterflies (addressing, exchanges, canonicalisation, the duplicated top layer), so its own
        # issue ceiling is lower: the rate kernel's instruction throughput divided by the kernel's instructions per row.
        tk = (traffic_rec or {}).get(r["kernel"], {})
        ipb = (traffic_rec or {}).get("rate_kernel_valu_insts_per_butterfly", {}).get(str(kind))
        if tk.get("valu_wave_insts_per_row") and ipb:
            wave_insts_s = rate * ipb / 64.0
            rows_issue = wave_insts_s / tk["valu_wave_insts_per_row"]
            r["valu_ceiling"].update({"valu_insts_per_butterfly": ipb, "kernel_valu_wave_insts_per_row": tk["valu_wave_insts_per_row"],
                                      "issue_ceiling_rows_per_s": rows_issue,
                                      "issue_ceiling_as_frac_of_hbm": rows_issue * 16 * n / 1e9 / HBM_PEAK_GBS})
            r["alu_ceiling_frac_butterflies_only"] = r["alu_ceiling_frac"]
            r["alu_ceiling_frac"] = r["frac"] / r["valu_ceiling"]["issue_ceiling_as_frac_of_hbm"]
        # SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES of the same passes: how full the vector ALUs are in CYCLES (the single-pass
        # kernels hold four waves per SIMD, so a wave can have a vector instruction executing in at most a quarter of its cycles).
        # Larger than alu_ceiling_frac by the clock: the rate kernel alone runs faster under the power limit than the transform.
        if tk.get("valu_active_share_of_wave_cycles"):
            r["valu_busy_frac"] = min(1.0, 4.0 * tk["valu_active_share_of_wave_cycles"])

    roof, traffic_rec = None, None
    tpath = os.path.join(ROOT, "profiles", "traffic.json")
    if prof and os.path.exists(tpath):
        try:
            rec = json.load(open(tpath))
            if rec.get("kernels_sha") == kernels_sha() and rec.get("config", 3) == args.config:
                traffic_rec = rec  # a record of another build or config is ignored (traffic stays null)
        except Exception:
            pass
    if prof:
        dominant = max(prof, key=lambda tkey: prof[tkey]["ms"])
        roof = ntt_roofline(dominant) if dominant in PASSES else (ntt_roofline("ntt_fwd_half") or ntt_roofline("ntt_fwd_pass"))
        if roof is None:
            roof = {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None}
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS
        # command, --measure-traffic; gfx950 correction applied): profiles/traffic.json records bytes per row for the
        # build it was measured on
        if traffic_rec and roof.get("rows_per_launch") and roof["kernel"] in traffic_rec:
            roof["traffic"] = traffic_rec[roof["kernel"]]["hbm_bytes_per_row_per_launch"] * roof["rows_per_launch"]
            roof["traffic_source"] = "profiles/traffic.json (%s)" % traffic_rec.get("measured", "rocprofv3 --pmc")
        valu_ceiling(roof)

    # ---- the pipeline-level roofline and the top kernels against their own algorithmic bytes
    pipeline, kernels = None, None
    if prof:
        ub = unit_bytes(cfg)
        per_gpu = value / world
        pipeline = {
            "unit": cfg["unit"].split("/")[0],
            "compulsory_bytes_per_unit": ub["compulsory"],
            "compulsory_frac_of_hbm": ub["compulsory"] * per_gpu / 1e9 / HBM_PEAK_GBS,
            "ntt_rows_per_unit": ub["ntt_rows"],
            "ntt_equivalent_bytes_per_unit": ub["ntt_equivalent"],
            "ntt_equivalent_frac_of_hbm": ub["ntt_equivalent"] * per_gpu / 1e9 / HBM_PEAK_GBS,
            "measured_hbm_bytes_per_unit": traffic_rec["step"]["hbm_bytes_per_unit"] if traffic_rec and "step" in traffic_rec else None,
        }
        if pipeline["measured_hbm_bytes_per_unit"]:
            pipeline["measured_frac_of_hbm"] = pipeline["measured_hbm_bytes_per_unit"] * per_gpu / 1e9 / HBM_PEAK_GBS
            pipeline["measured_over_compulsory"] = pipeline["measured_hbm_bytes_per_unit"] / ub["compulsory"]
        kernels = []
        units_timed = B * args.steps
        for tag, v in sorted(prof.items(), key=lambda kv: -kv[1]["ms"])[:5]:
            if tag in PASSES:
                alg = v["units"] * 16 * n / PASSES[tag]
            else:
                per_unit = kernel_bytes_per_unit(cfg, tag)
                alg = per_unit * units_timed if per_unit else None
            ent = {"kernel": tag, "ms_per_step": v["ms"] / args.steps, "launches_per_step": v["launches"] / args.steps,
                   "algorithmic_bytes_per_step": alg / args.steps if alg else None,
                   "frac_of_hbm": (alg / (v["ms"] / 1e3) / 1e9 / HBM_PEAK_GBS) if alg else None}
            if traffic_rec and tag in traffic_rec.get("per_kernel", {}):
                ent["measured_hbm_bytes_per_unit"] = traffic_rec["per_kernel"][tag]
            kernels.append(ent)

    # ---- NTT-only section: forward-NTT/s (the other half of the BASELINE metric), same primes, same device
    ntt = None
    if not args.stub and args.ntt_polys > 0:
        P = args.ntt_polys if cfg["logn"] < 16 else max(1, args.ntt_polys // 4)
        with torch.cuda.stream(w.stream):
            x = torch.empty((P, k, n), dtype=torch.int64, device=w.dev)
            fill_mod_rows(x, w.kmods[:k])
            w.ctx.ntt_negacyclic_harvey(x, P, k)
        barrier_sync()
        w.ctx.profile_enable(True)
        reps = 10
        t1 = time.perf_counter()
        for _ in range(reps):
            w.ctx.ntt_negacyclic_harvey(x, P, k)
        barrier_sync()
        ntt_dt = max_over_ranks(time.perf_counter() - t1)
        nprof = w.ctx.profile_fetch()
        w.ctx.profile_enable(False)
        ntt_rows = P * k * reps
        ntt_kernel_s = sum(v["ms"] for tkey, v in nprof.items() if tkey.startswith("ntt_fwd")) / 1e3
        ntt = {
            "forward_ntt_per_s": world * ntt_rows / ntt_dt,
            "forward_ntt_per_s_kernel_time": ntt_rows / ntt_kernel_s,
            "hbm_roofline_frac": (ntt_rows * 16 * n / ntt_kernel_s) / 1e9 / HBM_PEAK_GBS,
            "rows": P * k, "reps": reps, "n": n,
            "rows_note": "%d polynomials x the k = %d first-level primes of this config per launch (the key level has one "
                         "more prime; the rate is per row)" % (P, k),
        }
        if roof and roof.get("valu_ceiling"):
            vc = roof["valu_ceiling"]  # (the in-step launches' instruction count per row stands in for this launch's)
            ntt["alu_ceiling_frac"] = ntt["hbm_roofline_frac"] / vc.get("issue_ceiling_as_frac_of_hbm", vc["as_frac_of_hbm"])
        del x

    # ---- PCIe-inclusive rate of the same step (N = 1): host-pointer entries on separately allocated pageable ciphertexts
    pcie = None
    if not args.stub and world == 1 and args.pcie_pairs > 0:
        try:
            pcie = w.pcie_inclusive(args.pcie_pairs)
        except Exception as exc:  # an auxiliary section must never cost the run its line: report what failed instead
            pcie = {"error": "%s: %s" % (type(exc).__name__, exc)}
            try:
                w.ctx.synchronize()
            except Exception:
                pass

    # ---- the final gather (N > 1, or --force-dist): every rank's result slice to rank 0 over RCCL, timed on its own
    gather = gather_payload(w.result_slice(min(B, args.gather_cts)), cheap_digest, force=args.force_dist)
    key_digests = gather_digests(w.key_digest())  # the key must be the same on every rank
    rank_digests = gather_digests(cheap_digest(w.result_slice(min(B, 4))))

    # ---- checker leg: the timed path's output against the CPU oracle, then the CPU baseline (rank 0, N = 1)
    verified, items, cpu, bad_items = None, [], None, []
    if not args.stub and not (args.no_verify and (args.no_cpu_baseline or world > 1)):
        O, how = load_oracle()
        if not args.no_verify:
            items = w.items
            bad_items = w.verify(O)
            verified = all_ranks_true(not bad_items)
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(O, how, cfg, args.cpu_ops, w.elt, args.mode == "strict")
            cpu["gpu_over_cpu_%dthreads_measured" % cpu["cores"]] = value / cpu["value"]
            cpu["gpu_over_cpu_1thread"] = value / cpu["value_1thread"]
            cpu["gpu_over_cpu_all_physical_cores_projected"] = value / cpu["projected_all_physical_cores_linear"]

    if rank == 0:
        line = {
            "metric": "stub rank-logic rehearsal (NOT the engine)" if args.stub else (
                cfg["metric"].replace("bit-exact PARITY mode", "STRICT mode: the results that decrypt, SURVEY B.6")
                if args.mode == "strict" else cfg["metric"]),
            "value": value,
            "unit": cfg["unit"],
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": dt / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": "u64",
            "data": "synthetic",
            "config": {"workload": cfg["workload"], "baseline_config": args.config,
                       "ciphertexts_per_gpu": B, "global_batch": world * B, "mode": args.mode.upper(),
                       "parallelism": "dp%d (independent ciphertexts sharded, no data-path collective)" % world},
            "roofline": roof,
            "pipeline_roofline": pipeline,
            "kernels": kernels,
            "cpu_baseline": cpu,
            "ntt": ntt,
            "kernel_time_shares": shares,
            "pcie_inclusive": pcie,
            "verified_count": len(items),
            "verified_items": items,
            "verified_chunk_sizes": getattr(w, "chunks", []),
            "verified_vs_oracle": verified,
            "gather": gather,
            "rccl_ranks_seen": gather["ranks_seen"] if gather else (1 if world == 1 else 0),
            "dist_initialized": _dist_on(),
            "key_replicated": len(set(key_digests)) == 1,
            "rank_digests": ["%016x" % d for d in rank_digests],
        }
        if args.stub:
            line["stub"] = True
        print(json.dumps(line), flush=True)
    if _dist_on():
        dist.barrier()
        dist.destroy_process_group()
    if verified is False:
        raise SystemExit("bench.py: the timed path's output differs from the CPU oracle (items %s of the %d checked on rank %d)"
                         % (bad_items, len(items), rank))


KERNEL_TAGS = (("ntt_fwd_half_kernel", "ntt_fwd_half"), ("ntt_inv_half_kernel", "ntt_inv_half"), ("ntt_inv_top_kernel", "ntt_inv_top"),
               ("ntt_pass_kernel", "ntt_pass"), ("bfv_lift", "bfv_lift"), ("bfv_floor_sk", "bfv_floor_sk"),
               ("ks_mac", "ks_mac"), ("ks_moddown_bfv", "ks_moddown_bfv"), ("ks_moddown_post", "ks_moddown_post"),
               ("ks_moddown_pre", "ks_moddown_pre"), ("galois_kernel", "galois"), ("divround_bfv", "divround_bfv"),
               ("tensor_product", "tensor_product"), ("copy_rows", "copy_rows"))


def measure_traffic(args):
    """--measure-traffic: HBM bytes of the step from the PMC counters, collected as MI355X_MICROARCH.md prescribes
    (FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes over this very command at a reduced batch; values in
    KB; on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane read stream and is doubled). Per launch and row for the
    NTT kernels (roofline.traffic), per unit for every kernel and for the step as a whole (pipeline_roofline, kernels).
    Runs the passes as child processes before this process touches the GPU and records the result, with the identity of
    the kernel sources and the config, in profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    batch, steps = min(256, args.batch), 2
    totals = {}  # tag -> {"FETCH_SIZE": KB, "WRITE_SIZE": KB, "SQ_INSTS_VALU": wave instructions, "rows": rows, "launches": n}
    everything = {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0, "SQ_ACTIVE_INST_VALU": 0.0, "SQ_WAVE_CYCLES": 0.0}
    rate_kernel_insts = {}  # kind -> SQ_INSTS_VALU of the butterfly-rate kernel (bench.py's valu_ceiling runs it in the child too)
    # (SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES: the share of a resident wave's cycles with a vector instruction executing -- with
    #  the kernels' four waves per SIMD, four times that is how full the vector ALUs are, in cycles; optional: a pass that
    #  fails leaves the traffic figures alone)
    for counter in ("FETCH_SIZE", "WRITE_SIZE", "SQ_INSTS_VALU", "SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES"):
        optional = counter in ("SQ_ACTIVE_INST_VALU", "SQ_WAVE_CYCLES")
        d = tempfile.mkdtemp(prefix="sealhip_pmc_", dir="/tmp")
        # the step only (no NTT-only section): the same mix of launches that roofline.achieved is measured over
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--config", str(args.config), "--batch", str(batch), "--steps", str(steps),
               "--warmup", "0", "--no-cpu-baseline", "--no-verify", "--ntt-polys", "0", "--pcie-pairs", "0", "--mode", args.mode]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            if optional:
                shutil.rmtree(d, ignore_errors=True)
                continue
            return None
        for row in csv.DictReader(open(files[0])):
            name = row["Kernel_Name"]
            val = float(row["Counter_Value"])
            if "fill_mod" in name or "at::native" in name or "distribution" in name:
                continue  # torch's input fills are not part of the step
            if "butterfly_rate_kernel" in name:
                if counter == "SQ_INSTS_VALU":
                    kind = name.split("butterfly_rate_kernel<")[1].split(">")[0]
                    rate_kernel_insts[kind] = rate_kernel_insts.get(kind, 0.0) + val
                continue  # the ceiling microbenchmark is not part of the step either
            everything[counter] += val
            tag = next((t for key, t in KERNEL_TAGS if key in name), None)
            if tag is None:
                continue
            t = totals.setdefault(tag, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "SQ_INSTS_VALU": 0.0, "SQ_ACTIVE_INST_VALU": 0.0,
                                        "SQ_WAVE_CYCLES": 0.0, "rows": 0.0, "launches": 0})
            t[counter] += val
            if counter == "FETCH_SIZE":
                t["rows"] += int(row["Grid_Size"]) / int(row["Workgroup_Size"]) / 2  # half kernels: two workgroups per row
                t["launches"] += 1
        shutil.rmtree(d, ignore_errors=True)
    units = batch * steps
    rec = {"kernels_sha": kernels_sha(), "config": args.config,
           "measured": time.strftime("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %Y-%m-%d"),
           "command": "bench.py --config %d --batch %d --steps %d --warmup 0 --ntt-polys 0 (every launch of the step)"
                      % (args.config, batch, steps),
           "step": {"hbm_bytes_per_unit": (2 * everything["FETCH_SIZE"] + everything["WRITE_SIZE"]) * 1024 / units,
                    "fetch_kb_raw_total": everything["FETCH_SIZE"], "write_kb_total": everything["WRITE_SIZE"], "units": units},
           "per_kernel": {}}
    for tag, t in totals.items():
        rec["per_kernel"][tag] = (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / units
        if tag not in ("ntt_fwd_half", "ntt_inv_half") or not t["rows"]:
            continue
        # summed over every launch of the kernel in the step (the variants differ: gathered / in place, with or without
        # the fused tensor product), divided by the rows they transform
        rec[tag] = {"hbm_bytes_per_row_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / t["rows"],
                    "read_bytes_per_row": 2 * t["FETCH_SIZE"] * 1024 / t["rows"],
                    "write_bytes_per_row": t["WRITE_SIZE"] * 1024 / t["rows"],
                    "rows_total": t["rows"], "launches": t["launches"], "fetch_size_kb_raw_total": t["FETCH_SIZE"],
                    "write_size_kb_total": t["WRITE_SIZE"],
                    # wave-level vector-ALU instructions per transformed row (SQ_INSTS_VALU, its own --pmc pass)
                    "valu_wave_insts_per_row": t["SQ_INSTS_VALU"] / t["rows"]}
        if t["SQ_WAVE_CYCLES"] > 0 and t["SQ_ACTIVE_INST_VALU"] > 0:
            rec[tag]["valu_active_share_of_wave_cycles"] = t["SQ_ACTIVE_INST_VALU"] / t["SQ_WAVE_CYCLES"]
    # the butterfly-rate kernel under the same counter: wave-level VALU instructions per butterfly of the measured sequence
    # (csrc/ntt.hip ntt_butterfly_rate: a warm-up launch of iters / 8 and two of iters = 800, 2048 workgroups x 512 lanes x 32
    # butterflies per iteration; two calls -- the instance's sequence and the reference's -- when they differ)
    if rate_kernel_insts:
        per_call = (100 + 800 + 800) * 2048 * 512 * 32  # butterflies of one sealhip_debug_butterfly_rate call
        calls = {kd: max(1, round(v * 64 / per_call / 15)) for kd, v in rate_kernel_insts.items()}  # (~15 instructions each)
        rec["rate_kernel_valu_insts_per_butterfly"] = {kd: v * 64 / (per_call * calls[kd]) for kd, v in rate_kernel_insts.items()}
    json.dump(rec, open(os.path.join(ROOT, "profiles", "traffic.json"), "w"), indent=1)
    return rec


def kernels_sha():
    """Identity of the HIP sources a traffic record belongs to."""
    h = hashlib.sha1()
    d = os.path.join(ROOT, "gemini-seal_amd", "csrc")
    for f in sorted(os.listdir(d)):
        if f.endswith((".hip", ".hpp")):
            h.update(open(os.path.join(d, f), "rb").read())
    return h.hexdigest()[:16]


if __name__ == "__main__":
    main()
