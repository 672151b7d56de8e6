#!/usr/bin/env python3
"""bench.py -- headline benchmark of the sealhip engine.

Metric (BASELINE.json): ciphertext multiply+relinearize/s (and forward-NTT/s) at N=2^15, 8 RNS primes.
A step = Evaluator::multiply + Evaluator::relinearize over one batch of independent synthetic BFV
ciphertexts that are already resident in HBM (config 3 of BASELINE.json: N=2^15, {55}x8 primes, k=7,
|Bsk|=8, 7 key digits, t=786433, PARITY mode = bit-exact with the reference).

    python bench.py --gpus N --steps K --warmup W [--batch B]

N > 1: one rank per GPU (torch.distributed, backend nccl = RCCL). When no launcher has started the ranks
(WORLD_SIZE unset) this process starts them itself as a child `python -m torch.distributed.run` -- before it
has touched the GPU -- and relays rank 0's JSON line. The batch of independent ciphertexts is sharded
contiguously over ranks with NO data-path collective (weak scaling: every rank owns `--batch` ciphertexts).
Collectives: the timing barrier / max-over-ranks, and after the timed region the final gather SURVEY 8(e) names
(every rank's output slice to rank 0 over RCCL/xGMI), timed separately and checked by digest.

The timed path verifies itself: after the timed region ciphertext pairs 0, B/2 (inside the middle arena chunk)
and B-1 are pulled back to the host and compared word for word with the CPU oracle's multiply+relinearize of
the same inputs (`verified_items`); a mismatch fails the run.

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


def gather_payload(payload, digest_fn):
    """The final gather of SURVEY 8(e): every rank's output slice to rank 0 (RCCL send/recv over xGMI under the nccl
    backend), timed on its own, then checked on rank 0 against the digests the ranks computed locally.
    Returns None for a single rank."""
    if not _dist_on() or dist.get_world_size() == 1:
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
    return {"bytes_per_rank": nbytes, "seconds": dt, "GBps_into_root": (world - 1) * nbytes / dt / 1e9,
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


def verify_against_oracle(O, key, a, b, out, items):
    """multiply+relinearize of the given ciphertext pairs on the CPU oracle vs the words the engine produced."""
    L = O.lib()
    n = 1 << LOGN
    kmods = O.coeff_modulus_create(n, BITS)
    assert kmods == CFG3_PRIMES
    ref = O.RefContext(1, LOGN, kmods, nsp=NSP, t=PLAIN_T)
    k = ref.k_first
    hkey = host_u64(key)
    keys = (C.c_void_p * 1)(hkey.ctypes.data)
    ok = []
    for i in items:
        ha, hb, got = host_u64(a[i]), host_u64(b[i]), host_u64(out[i])
        exp = np.zeros((3, k, n), dtype=np.uint64)
        assert L.ref_bfv_multiply(C.byref(ref.c), k, O.ptr(ha), 2, O.ptr(hb), 2, O.ptr(exp)) == 0
        assert L.ref_relinearize(C.byref(ref.c), k, O.ptr(exp), 3, keys) == 0
        ok.append(bool(np.array_equal(got, exp)))
    return ok


def cpu_baseline(O, how, total_ops):
    """Times the CPU oracle (oracle/sealref.c, digest-identical to the reference) on a bounded sample of the SAME
    workload: one worker thread per CPU this process may use -- min(logical CPUs, scheduler affinity, cgroup CPU quota);
    threads beyond the quota only get throttled -- each doing multiply+relinearize on its own ciphertexts (the
    reference is single-threaded per call and thread-safe across calls). The host's CPU model and core counts are
    reported next to it, with the linear projection of the 1-thread rate to every physical core of the host (an upper
    bound: it assumes perfect scaling), because a GPU box hands its container only a share of the host's cores."""
    import math
    from concurrent.futures import ThreadPoolExecutor

    L = O.lib()
    info = cpu_info()
    usable = [c for c in (info["logical_cpus"], info["affinity_cpus"],
                          math.ceil(info["cgroup_cpu_quota"]) if info["cgroup_cpu_quota"] else None) if c]
    threads = max(1, min(usable) if usable else 1)
    per_thread = max(1, int(round(total_ops / threads)))
    n = 1 << LOGN
    kmods = O.coeff_modulus_create(n, BITS)
    ref = O.RefContext(1, LOGN, kmods, nsp=NSP, t=PLAIN_T)
    k = ref.k_first
    ref.rns_tool(k)  # build shared constants before the threads start
    rng = np.random.default_rng(1)

    def rows(mods):
        return np.stack([rng.integers(0, p, size=n, dtype=np.uint64) for p in mods])

    key = np.stack([rows(kmods * 2).reshape(2, len(kmods), n) for _ in range(k)])
    work = [(rows(kmods[:k] * 2), rows(kmods[:k] * 2), np.zeros((3, k, n), dtype=np.uint64)) for _ in range(threads)]
    keys = (C.c_void_p * 1)(key.ctypes.data)

    def run(item, reps=per_thread):
        a, b, out = item
        for _ in range(reps):
            assert L.ref_bfv_multiply(C.byref(ref.c), k, O.ptr(a), 2, O.ptr(b), 2, O.ptr(out)) == 0
            assert L.ref_relinearize(C.byref(ref.c), k, O.ptr(out), 3, keys) == 0

    run(work[0], 1)  # page everything in
    t1 = time.perf_counter()
    one_reps = 4
    run(work[0], one_reps)
    one_thread = one_reps / (time.perf_counter() - t1)
    t0 = time.perf_counter()
    with ThreadPoolExecutor(max_workers=threads) as ex:
        list(ex.map(run, work))
    dt = time.perf_counter() - t0
    # forward NTT/s on one core, same primes
    x = rows(kmods[:k])
    t1 = time.perf_counter()
    reps = 8
    for _ in range(reps):
        for i in range(k):
            L.ref_ntt_forward(O.ptr(x[i]), ref.tables(i), 0)
    ntt_s = reps * k / (time.perf_counter() - t1)
    out = {
        "value": threads * per_thread / dt,
        "unit": "ct_mul_relin/s",
        "cores": threads,
        "kind": "port",
        "sample": "%d threads (one per CPU usable by this container) x %d BFV multiply+relinearize at N=2^15, 8 primes "
                  "(same workload, %d ciphertext pairs)" % (threads, per_thread, threads * per_thread),
        "seconds": dt,
        "value_1thread": one_thread,
        "forward_ntt_per_s_1core": ntt_s,
        "build": how,
    }
    out.update(info)
    cores = info["physical_cores"] or info["logical_cpus"] or threads
    out["projected_all_physical_cores_linear"] = one_thread * cores
    return out


# ------------------------------------------------------------------ workloads
class EngineWorkload:
    """config 3 on one MI355X through the C ABI (ctypes mirror of the Evaluator interface)."""

    def __init__(self, args, rank, local_rank):
        import sealhip as S

        if not torch.cuda.is_available() or S.num_devices() < 1:
            raise SystemExit("bench.py needs a HIP device: the engine has no CPU fallback")
        torch.cuda.set_device(local_rank)
        self.dev = torch.device("cuda", local_rank)
        # one explicit stream for torch's fills/copies AND the engine's launches: everything below is ordered on it
        self.stream = torch.cuda.Stream(device=self.dev)
        self.n, self.kmods = 1 << LOGN, CFG3_PRIMES
        self.ctx = S.Context(S.SCHEME_BFV, LOGN, self.kmods, NSP, PLAIN_T, device=local_rank)
        self.ctx.set_stream(self.stream.cuda_stream)
        self.ev = S.Evaluator(self.ctx)
        self.S = S
        self.k, self.nk, self.B = self.ctx.k_first, len(self.kmods), args.batch
        k, n, B, dev = self.k, self.n, self.B, self.dev
        with torch.cuda.stream(self.stream):
            torch.manual_seed(1234 + rank)  # every rank owns different ciphertexts
            self.a = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
            self.b = torch.empty((B, 2, k, n), dtype=torch.int64, device=dev)
            self.out = torch.empty((B, 3, k, n), dtype=torch.int64, device=dev)
            self.key = torch.empty((k, 2, self.nk, n), dtype=torch.int64, device=dev)
            fill_mod_rows(self.a, self.kmods[:k])
            fill_mod_rows(self.b, self.kmods[:k])
            torch.manual_seed(99)  # the relinearisation key is replicated on every GPU
            fill_mod_rows(self.key, self.kmods)
            self.rk = S.KSwitchKeys(self.ctx, self.key, n_digits=k, from_host=False)

    def step(self):
        self.ev.multiply(self.a, 2, self.b, 2, self.k, self.B, self.out)
        self.ev.relinearize_inplace(self.out, 3, self.k, self.B, [self.rk])

    def finish(self):
        self.ctx.synchronize()  # also surfaces a device-side failure of any launch (sticky flag)

    def result_slice(self, count):
        with torch.cuda.stream(self.stream):
            res = self.out[:count, :2].contiguous()  # the size-2 results, compacted (what a caller gets back)
        self.stream.synchronize()  # the callers (digests, RCCL gather) work on other streams
        return res

    def key_digest(self):
        with torch.cuda.stream(self.stream):
            return cheap_digest(self.key[0, 0, :1])


class StubWorkload:
    """CPU stand-in with the same rank logic (seeds, replicated key, step, result slice) for the gloo tests of the
    launcher: NOT the engine and never timed as such (`"stub": true` in the line)."""

    def __init__(self, args, rank, local_rank):
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
        self.out[:, :2] = (self.a * self.b + self.key[0, 0, 0, 0]) % self.kmods[0]

    def finish(self):
        pass

    def result_slice(self, count):
        return self.out[:count, :2].contiguous()

    def key_digest(self):
        return cheap_digest(self.key[0, 0, :1])


# ------------------------------------------------------------------ main
def parse_args(argv=None):
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=3)
    ap.add_argument("--warmup", type=int, default=1)
    ap.add_argument("--batch", type=int, default=int(os.environ.get("SEALHIP_BENCH_BATCH", "4096")),
                    help="independent ciphertext pairs per GPU (BASELINE config 3: 4096)")
    ap.add_argument("--ntt-polys", type=int, default=4096,
                    help="polynomials (x7 rows) in the NTT-only section; 4096 = the batch of the step (the rate grows "
                         "with the launch: 31 %% of the roofline at 7 k rows, 36-38 %% at 29-57 k, DESIGN.md section 6)")
    ap.add_argument("--gather-cts", type=int, default=512,
                    help="size-2 result ciphertexts per rank in the final gather to rank 0 (N > 1 only; 3.67 MB each)")
    ap.add_argument("--cpu-ops", type=int, default=240,
                    help="multiply+relinearize operations of the CPU baseline sample (about 85 ms of one core each)")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-verify", action="store_true", help="skip the oracle check of the timed path's output")
    ap.add_argument("--measure-traffic", action="store_true",
                    help="first collect the HBM PMC counters of the NTT kernels (two rocprofv3 --pmc child runs of this "
                         "command at batch 256) and record them in profiles/traffic.json; roofline.traffic then comes "
                         "from this run's own measurement")
    ap.add_argument("--stub", action="store_true", help="CPU stand-in workload + gloo (tests of the rank logic only)")
    return ap.parse_args(argv)


def main(argv=None):
    args = parse_args(argv)
    if args.gpus < 1:
        raise SystemExit("--gpus must be at least 1")
    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args))  # nothing above has touched the GPU
    if args.measure_traffic and args.gpus == 1 and not args.stub:
        measure_traffic(args)  # child processes; this one has not touched the GPU yet
    world = int(os.environ.get("WORLD_SIZE", "1"))
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    if args.gpus != world:
        raise SystemExit("--gpus %d does not match the %d launched ranks (WORLD_SIZE)" % (args.gpus, world))
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.stub:
            dist.init_process_group("gloo")
        else:
            torch.cuda.set_device(local_rank)
            dist.init_process_group("nccl", device_id=torch.device("cuda", local_rank))
    w = (StubWorkload if args.stub else EngineWorkload)(args, rank, local_rank)
    B, k, n = w.B, w.k, w.n
    lo, hi = shard_range(world * B, rank, world)  # this rank's slice of the global batch (weak scaling: B each)
    assert hi - lo == B

    for _ in range(args.warmup):
        w.step()
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

    # launches per full row transform at N=2^15: the single-pass kernels complete a transform per launch, the tiled
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

    roof = None
    if prof:
        dominant = max(prof, key=lambda tkey: prof[tkey]["ms"])
        roof = ntt_roofline(dominant) if dominant in PASSES else (ntt_roofline("ntt_fwd_half") or ntt_roofline("ntt_fwd_pass"))
        if roof is None:
            roof = {"kernel": dominant, "bound": "hbm", "achieved": None, "peak": HBM_PEAK_GBS, "unit": "GB/s",
                    "frac": None, "traffic": None}
        # HBM bytes per launch from the PMC counters (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes over THIS
        # command, tools/measure_traffic.sh; gfx950 correction applied): profiles/traffic.json records bytes per row for
        # the build it was measured on; a record of another build is ignored (traffic stays null)
        tpath = os.path.join(ROOT, "profiles", "traffic.json")
        if os.path.exists(tpath) and roof.get("rows_per_launch"):
            try:
                rec = json.load(open(tpath))
                if rec.get("kernels_sha") == kernels_sha():
                    roof["traffic"] = rec[roof["kernel"]]["hbm_bytes_per_row_per_launch"] * roof["rows_per_launch"]
                    roof["traffic_source"] = "profiles/traffic.json (%s)" % rec.get("measured", "rocprofv3 --pmc")
            except Exception:
                pass

    # ---- NTT-only section: forward-NTT/s (the other half of the BASELINE metric), same primes, same device
    ntt = None
    if not args.stub and args.ntt_polys > 0:
        P = args.ntt_polys
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
            "rows": P * k, "reps": reps,
        }
        del x

    # ---- the final gather (N > 1): every rank's result slice to rank 0 over RCCL, timed on its own
    gather = gather_payload(w.result_slice(min(B, args.gather_cts)), cheap_digest)
    key_digests = gather_digests(w.key_digest())  # the key must be the same on every rank
    rank_digests = gather_digests(cheap_digest(w.result_slice(min(B, 4))))

    # ---- checker leg: the timed path's output against the CPU oracle, then the CPU baseline (rank 0, N = 1)
    verified, items, cpu = None, [], None
    if not args.stub and not (args.no_verify and (args.no_cpu_baseline or world > 1)):
        O, how = load_oracle()
        if not args.no_verify:
            items = sorted({0, B // 2, B - 1})
            with torch.cuda.stream(w.stream):
                oks = verify_against_oracle(O, w.key, w.a, w.b, w.out, items)
            verified = all_ranks_true(all(oks))
        if rank == 0 and world == 1 and not args.no_cpu_baseline:
            cpu = cpu_baseline(O, how, args.cpu_ops)
            cpu["gpu_over_cpu_%dthreads_measured" % cpu["cores"]] = value / cpu["value"]
            cpu["gpu_over_cpu_1thread"] = value / cpu["value_1thread"]
            cpu["gpu_over_cpu_all_physical_cores_projected"] = value / cpu["projected_all_physical_cores_linear"]

    if rank == 0:
        line = {
            "metric": "stub rank-logic rehearsal (NOT the engine)" if args.stub else
                      "ciphertext multiply+relinearize/s (BFV N=2^15, 8 primes, bit-exact PARITY mode)",
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
            "verified_items": items,
            "verified_vs_oracle": verified,
            "gather": gather,
            "rccl_ranks_seen": gather["ranks_seen"] if gather else (1 if world == 1 else 0),
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
        raise SystemExit("bench.py: the timed path's output differs from the CPU oracle (items %s)" % items)


def measure_traffic(args):
    """--measure-traffic: HBM bytes per launch of the NTT kernels from the PMC counters, collected as
    MI355X_MICROARCH.md prescribes (FETCH_SIZE and WRITE_SIZE in their own rocprofv3 --pmc passes over this very
    command at a reduced batch; values in KB; on gfx950 FETCH_SIZE reports half the bytes of a 16 B/lane read stream and
    is doubled). Runs the passes as child processes before this process touches the GPU and records the result, with
    the identity of the kernel sources, in profiles/traffic.json."""
    import csv
    import glob
    import shutil
    import tempfile

    exe = shutil.which("rocprofv3")
    if not exe:
        return None
    totals = {}  # tag -> {"FETCH_SIZE": KB, "WRITE_SIZE": KB, "rows": rows, "launches": n}
    for counter in ("FETCH_SIZE", "WRITE_SIZE"):
        d = tempfile.mkdtemp(prefix="sealhip_pmc_", dir="/tmp")
        # the multiply+relinearize step only (no NTT-only section): the same mix of forward / inverse launches that
        # roofline.achieved is measured over
        cmd = [exe, "--pmc", counter, "--output-format", "csv", "-d", d, "-o", "p", "--", sys.executable,
               os.path.abspath(__file__), "--batch", "256", "--steps", "2", "--warmup", "0", "--no-cpu-baseline",
               "--no-verify", "--ntt-polys", "0"]
        env = dict(os.environ, TMPDIR="/tmp")
        r = subprocess.run(cmd, cwd="/tmp", env=env, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL, timeout=900)
        files = glob.glob(os.path.join(d, "**", "*counter_collection.csv"), recursive=True)
        if r.returncode != 0 or not files:
            return None
        for row in csv.DictReader(open(files[0])):
            name = row["Kernel_Name"]
            if "ntt_fwd_half_kernel" in name:
                tag = "ntt_fwd_half"
            elif "ntt_inv_half_kernel" in name:
                tag = "ntt_inv_half"
            else:
                continue
            t = totals.setdefault(tag, {"FETCH_SIZE": 0.0, "WRITE_SIZE": 0.0, "rows": 0.0, "launches": 0})
            t[counter] += float(row["Counter_Value"])
            if counter == "FETCH_SIZE":
                t["rows"] += int(row["Grid_Size"]) / int(row["Workgroup_Size"]) / 2  # two workgroups per row
                t["launches"] += 1
        shutil.rmtree(d, ignore_errors=True)
    rec = {"kernels_sha": kernels_sha(), "measured": time.strftime("rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE, %Y-%m-%d"),
           "command": "bench.py --batch 256 --steps 2 --warmup 0 --ntt-polys 0 (every launch of the kernel in the step)"}
    for tag, t in totals.items():
        if not t["rows"]:
            continue
        # summed over every launch of the kernel in the step (the variants differ: gathered / in place, with or without
        # the fused tensor product), divided by the rows they transform
        rec[tag] = {"hbm_bytes_per_row_per_launch": (2 * t["FETCH_SIZE"] + t["WRITE_SIZE"]) * 1024 / t["rows"],
                    "rows_total": t["rows"], "launches": t["launches"], "fetch_size_kb_raw_total": t["FETCH_SIZE"],
                    "write_size_kb_total": t["WRITE_SIZE"]}
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
