"""CPU-only checks of the product's host side: the C-ABI library loads and exports every symbol that
include/sealhip.h declares, the tables/constants it regenerates equal the oracle's (and therefore the
reference's), errors map to the reference's HRESULT convention, and nothing computes without a GPU."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import oracle_lib as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope="module")
def S():
    import sealhip

    return sealhip


def test_library_exports_every_declared_symbol(S):
    header = open(os.path.join(ROOT, "include", "sealhip.h")).read()
    declared = set(re.findall(r"\b(sealhip_[a-z0-9_]+)\s*\(", header))
    assert declared, "no declarations found"
    lib = C.CDLL(S.LIB_PATH)
    for name in sorted(declared):
        assert hasattr(lib, name), "missing export: " + name
    assert declared == set(S.SYMBOLS), declared ^ set(S.SYMBOLS)


@pytest.mark.parametrize("logn,bits", [(3, [20, 21]), (6, [30] * 4), (12, [36, 36, 37]), (15, [55, 50])])
def test_ntt_tables_equal_oracle(S, logn, bits):
    n = 1 << logn
    mods = O.coeff_modulus_create(n, bits)
    ctx = S.Context(S.SCHEME_BFV, logn, mods, 1, 65537, device=-1)
    aux = O.get_primes(n, 60, len(mods) + 3)
    names = ["root_powers", "scaled_root_powers", "inv_root_powers", "scaled_inv_root_powers"]
    for idx, p in enumerate(mods + aux):
        if idx == len(mods) + 1:
            with pytest.raises(ValueError):
                ctx.debug_ntt_table(idx, 0)  # gamma has no tables
            continue
        t = O.Tables(logn, p)
        for kind, name in enumerate(names):
            assert np.array_equal(ctx.debug_ntt_table(idx, kind), t.arr(name)), (p, name)


@pytest.mark.parametrize("logn,bits,t", [(6, [30] * 5, 65537), (12, [36, 36, 37], 786433), (9, [59, 59, 59, 59, 40], (1 << 58) + 1)])
def test_rns_constants_equal_oracle(S, logn, bits, t):
    n = 1 << logn
    mods = O.coeff_modulus_create(n, bits)
    ctx = S.Context(S.SCHEME_BFV, logn, mods, 1, t, device=-1)
    ref = O.RefContext(1, logn, mods, 1, t)
    for k in range(1, len(mods)):
        rt = ref.rns_tool(k).contents
        nb, B = rt.Bsk_size, rt.B_size
        arr = lambda p, cnt: [int(p[i]) for i in range(cnt)]
        assert [int(v) for v in ctx.debug_rns_constants(k, 0)] == [int(rt.Bsk[i].value) for i in range(nb)]
        assert [int(v) for v in ctx.debug_rns_constants(k, 1)] == arr(rt.inv_prod_q_mod_Bsk, nb)
        assert [int(v) for v in ctx.debug_rns_constants(k, 2)] == arr(rt.prod_q_mod_Bsk, nb)
        assert [int(v) for v in ctx.debug_rns_constants(k, 3)] == arr(rt.inv_m_tilde_mod_Bsk, nb)
        assert [int(v) for v in ctx.debug_rns_constants(k, 4)] == arr(rt.prod_B_mod_q, k)
        if k > 1:
            assert [int(v) for v in ctx.debug_rns_constants(k, 5)] == arr(rt.inv_q_last_mod_q, k - 1)
        assert [int(v) for v in ctx.debug_rns_constants(k, 6)] == [
            rt.inv_prod_q_mod_m_tilde, rt.inv_prod_B_mod_m_sk, rt.m_sk.value, rt.gamma.value]
        assert [int(v) for v in ctx.debug_rns_constants(k, 7)] == arr(rt.q_to_Bsk.matrix, nb * k)
        assert [int(v) for v in ctx.debug_rns_constants(k, 8)] == arr(rt.B_to_q.matrix, k * B)
        assert [int(v) for v in ctx.debug_rns_constants(k, 9)] == arr(rt.q_to_Bsk.inv_punct, k)
        assert [int(v) for v in ctx.debug_rns_constants(k, 10)] == arr(rt.B_to_q.inv_punct, B)
        assert [int(v) for v in ctx.debug_rns_constants(k, 11)] == arr(rt.q_to_m_tilde.matrix, k)
        assert [int(v) for v in ctx.debug_rns_constants(k, 12)] == arr(rt.B_to_m_sk.matrix, B)
        assert ctx.bsk_size(k) == nb


def test_no_cpu_fallback_and_error_mapping(S):
    mods = O.coeff_modulus_create(64, [30, 30, 30])
    ctx = S.Context(S.SCHEME_CKKS, 6, mods, 1, 0, device=-1)
    assert ctx.k_first == 2
    # a host-only context never computes: std::logic_error -> COR_E_INVALIDOPERATION
    with pytest.raises(S.LogicError):
        ctx.ntt_negacyclic_harvey(0x1000, 1, 2)
    with pytest.raises(S.LogicError):
        S.Evaluator(ctx).multiply(0x1000, 2, 0x2000, 2, 2, 1, 0x3000)
    with pytest.raises(S.LogicError):
        ctx.alloc(16)
    # (pinning host memory for the *_host entries needs a device too; a null pointer is E_POINTER first)
    buf = np.zeros(64, dtype=np.uint64)
    with pytest.raises(S.LogicError):
        S.Evaluator(ctx).host_register(buf)
    with pytest.raises(S.LogicError):
        S.Evaluator(ctx).host_unregister(buf)
    assert (S.lib().sealhip_host_register(ctx.handle, None, 64) & 0xFFFFFFFF) == S.E_POINTER
    # E_POINTER before anything else, like the reference's IfNullRet
    with pytest.raises(TypeError):
        ctx.ntt_negacyclic_harvey(0, 1, 2)
    # invalid parameters -> E_INVALIDARG (std::invalid_argument)
    for bad in (
        dict(scheme=3, log_n=6, key_moduli=mods),
        dict(scheme=S.SCHEME_CKKS, log_n=2, key_moduli=mods),
        dict(scheme=S.SCHEME_CKKS, log_n=17, key_moduli=mods),
        dict(scheme=S.SCHEME_CKKS, log_n=6, key_moduli=mods, n_special_primes=3),
        dict(scheme=S.SCHEME_CKKS, log_n=6, key_moduli=[mods[0], mods[0], mods[1]]),
        dict(scheme=S.SCHEME_CKKS, log_n=6, key_moduli=[97, 193]),  # 193 = 1 mod 128 but 97 is not
        dict(scheme=S.SCHEME_BFV, log_n=6, key_moduli=mods, plain_modulus=0),
    ):
        with pytest.raises(ValueError):
            S.Context(device=-1, **bad)
    if S.num_devices() == 0:
        with pytest.raises(RuntimeError):
            S.Context(S.SCHEME_CKKS, 6, mods, 1, 0, device=0)  # no HIP device -> E_UNEXPECTED, never a CPU path


def test_galois_elt_from_step_and_naf(S):
    mods = O.coeff_modulus_create(64, [30, 30])
    ctx = S.Context(S.SCHEME_CKKS, 6, mods, 1, 0, device=-1)
    L = O.lib()
    for step in range(-31, 32):
        assert ctx.galois_elt_from_step(step) == L.ref_galois_elt_from_step(64, step, None)
    with pytest.raises(ValueError):
        ctx.galois_elt_from_step(32)
    # util/numth.h:22-42
    assert S._naf(3) == [-1, 4] and S._naf(-3) == [1, -4] and S._naf(8) == [8] and S._naf(7) == [-1, 8]


def test_multi_rank_sharding_plan_gloo():
    """bench.py shards the batch of independent ciphertexts contiguously over ranks with no data-path
    collective; the N>1 path is exercised on CPU with world_size 2 over gloo."""
    import subprocess
    import sys

    script = os.path.join(ROOT, "tests", "_gloo_shard_worker.py")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
         "127.0.0.1", "--master-port", "29571", script],
        capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "SHARD_OK" in out.stdout


def test_latency_mode_digit_sharding_gloo():
    """SURVEY 8(e) latency mode on CPU with two gloo ranks: the digits of one key switch sharded contiguously over the ranks,
    partial inner products (the oracle's split of evaluator.cpp:2259-2368) summed by all_reduce(SUM) on 64-bit words, the
    rest of the key switch on the sum == the unsplit key switch, for CKKS / BFV and 1, 2, 3 special primes."""
    import subprocess
    import sys

    script = os.path.join(ROOT, "tests", "_gloo_latency_worker.py")
    out = subprocess.run(
        [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=2", "--master-addr",
         "127.0.0.1", "--master-port", "29575", script],
        capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stdout + out.stderr
    assert "LATENCY_OK" in out.stdout


def test_bench_spawns_its_own_ranks_and_gathers_gloo():
    """`python bench.py --gpus 2` with no launcher around it must start its two ranks itself (a child
    torch.distributed.run, never an exec), run the per-rank setup (rank-dependent ciphertext seed, replicated key,
    contiguous shard of the global batch), the barrier / max-over-ranks timing and the final payload gather to rank 0,
    and relay ONE JSON line. The stub workload stands in for the engine (no GPU here); backend gloo."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--steps", "2",
                          "--warmup", "1", "--batch", "6", "--gather-cts", "4"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["stub"] is True and line["n_gpus"] == 2 and line["scaling"] == "weak"
    assert line["config"]["global_batch"] == 12 and line["config"]["ciphertexts_per_gpu"] == 6
    assert line["rccl_ranks_seen"] == 2 and line["gather"]["ranks_seen"] == 2  # both slices arrived intact at rank 0
    assert line["gather"]["bytes_per_rank"] == 4 * 2 * 2 * 64 * 8
    assert line["key_replicated"] is True
    assert len(set(line["rank_digests"])) == 2  # every rank worked on its own ciphertexts
    # a launcher that started a different number of ranks is an error message, not an assert
    bad = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "3", "--stub"],
                         capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE="1", RANK="0"))
    assert bad.returncode != 0 and "does not match" in bad.stderr


@pytest.mark.parametrize("config", [4, 5])
def test_bench_other_baseline_configs_multi_rank_gloo(config):
    """`bench.py --config 4 / 5` (BASELINE's 8-GPU lines: rotate_vector over sharded ciphertexts with the replicated
    Galois key, evaluator.h:1201-1211; multiply + relinearize + mod_switch_to_next, evaluator.cpp:996-1036) go through
    the same launcher, sharding, timing and gather code as config 3: rehearsed with two gloo ranks and the stub step."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "2", "--stub", "--config", str(config),
                          "--steps", "2", "--warmup", "1", "--batch", "5", "--gather-cts", "3"],
                         capture_output=True, text=True, timeout=300, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 2 and line["config"]["baseline_config"] == config and line["config"]["global_batch"] == 10
    assert ("config %d" % config) in line["config"]["workload"]
    assert line["unit"] == {4: "rotate_vector/s", 5: "pipeline/s"}[config]
    assert line["rccl_ranks_seen"] == 2 and line["gather"]["ranks_seen"] == 2 and line["key_replicated"] is True
    assert len(set(line["rank_digests"])) == 2


@pytest.mark.parametrize("config", [3, 4, 5])
def test_bench_eight_rank_shape_rehearsal_gloo(config):
    """The shape the driver's SCALE run has, rehearsed on CPU before the first hardware run: EIGHT
    ranks started by bench.py itself as one `torch.distributed.run` child, shard_range(8 B, r, 8), rank-dependent seeds, the
    replicated key checked by digest, eight result slices of the default 512 ciphertexts each gathered into rank 0
    (scaled-down rows: the stub's ring), one JSON line with n_gpus 8 and rccl_ranks_seen 8. Backend gloo, stub step."""
    import json
    import subprocess
    import sys

    sys.path.insert(0, ROOT)
    import bench

    B = 520  # more than the 512 gathered ciphertexts, not a multiple of anything convenient
    for r in range(8):
        lo, hi = bench.shard_range(8 * B, r, 8)
        assert (lo, hi) == (r * B, (r + 1) * B)
    assert [bench.shard_range(8195, r, 8) for r in range(8)][-1][1] == 8195  # ragged totals are covered without gaps
    assert sum(hi - lo for lo, hi in (bench.shard_range(8195, r, 8) for r in range(8))) == 8195
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", "8", "--stub", "--config", str(config),
                          "--steps", "2", "--warmup", "1", "--batch", str(B)],
                         capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, out.stdout + out.stderr
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith("{")]
    assert len(lines) == 1, out.stdout
    line = json.loads(lines[0])
    assert line["n_gpus"] == 8 and line["scaling"] == "weak" and line["config"]["baseline_config"] == config
    assert line["config"]["global_batch"] == 8 * B and line["config"]["ciphertexts_per_gpu"] == B
    assert line["rccl_ranks_seen"] == 8 and line["gather"]["ranks_seen"] == 8  # eight slices arrived intact at rank 0
    assert line["gather"]["bytes_per_rank"] == 512 * 2 * 2 * 64 * 8  # the default --gather-cts, the stub's row length
    assert line["key_replicated"] is True and len(set(line["rank_digests"])) == 8
    assert line["dist_initialized"] is True


def test_bench_refuses_more_gpus_than_the_node_has():
    """`bench.py --gpus N` on a node with fewer devices fails fast with a clear message, before any rank is started."""
    import subprocess
    import sys

    import torch

    n = torch.cuda.device_count() + 1
    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK")}
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gpus", str(n)], capture_output=True, text=True,
                         timeout=300, env=env)
    assert out.returncode != 0 and "HIP device" in out.stderr and "nothing was launched" in out.stderr, out.stderr


def test_bench_force_dist_runs_the_collectives_with_one_rank_gloo():
    """--force-dist: a single rank still initialises the process group and runs barrier / all_reduce / all_gather and the
    final gather (to itself), so `rccl_ranks_seen` comes from the collective and not from the world == 1 shortcut. This is
    the CPU twin (gloo) of the GPU test that runs the same command on the nccl backend, with and without a launcher."""
    import json
    import subprocess
    import sys

    env = {k: v for k, v in os.environ.items() if k not in ("WORLD_SIZE", "RANK", "LOCAL_RANK", "MASTER_PORT")}
    base = [os.path.join(ROOT, "bench.py"), "--gpus", "1", "--stub", "--force-dist", "--steps", "1", "--batch", "4"]
    launcher = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node=1", "--master-addr",
                "127.0.0.1", "--master-port", "29573"]
    for cmd in ([sys.executable] + base, launcher + base):
        out = subprocess.run(cmd, capture_output=True, text=True, timeout=300, env=env)
        assert out.returncode == 0, out.stdout + out.stderr
        line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
        assert line["dist_initialized"] is True and line["n_gpus"] == 1
        assert line["gather"]["backend"] == "gloo" and line["gather"]["ranks_seen"] == 1 and line["rccl_ranks_seen"] == 1
    # without the flag the one-rank run does not touch torch.distributed at all
    out = subprocess.run([sys.executable] + [a for a in base if a != "--force-dist"], capture_output=True, text=True,
                         timeout=300, env=env)
    line = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert line["dist_initialized"] is False and line["gather"] is None and line["rccl_ranks_seen"] == 1


def _build_adapter(tmp_path):
    import subprocess

    exe = str(tmp_path / "host_adapter_check")
    libdir = os.path.join(ROOT, "gemini-seal_amd", "lib")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "host_adapter_check.cpp"),
                           "-L" + libdir, "-lsealhip", "-Wl,-rpath," + libdir])
    return exe


def test_cpp_host_adapter_compiles_and_links(tmp_path):
    """gemini-seal_amd/host/evaluator.hpp (the seal::Evaluator-shaped C++ adapter) builds against the ABI."""
    import subprocess

    out = subprocess.run([_build_adapter(tmp_path)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "host-only context ok" in out.stdout, out.stdout + out.stderr


def test_shortcut_bounds_are_proved_not_sampled(tmp_path):
    """csrc/ntt_bounds.hpp: every admission predicate of the launchers (lazy-sum inverse per shape, kNttAnyRep / kNttApprox /
    unreduced mod-up, fused tensor product, FP64 forward and inverse schedules) against its worst-case magnitude recurrence:
    prime sizes 20..61 bits x log n 14..16 x every schedule, admitted => below 2^64 / 2^53, one bit more => overflow; the
    round-2 whole-row bug (half-row bound on the larger shape) and round 2's six-layer FP64 spans both fail it; the FP64
    product bound and the recurrences are checked against bit-level executions (tests/bounds_check.cpp).
    Invariant: native/src/seal/util/defines.h:52-53, butterflies util/ntt.cpp:245-281."""
    import subprocess

    exe = str(tmp_path / "bounds_check")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-ffp-contract=off", "-o", exe,
                           os.path.join(ROOT, "tests", "bounds_check.cpp")])
    out = subprocess.run([exe], capture_output=True, text=True, timeout=300)
    assert out.returncode == 0 and "bounds_check: OK" in out.stdout, out.stdout + out.stderr


# ---------------------------------------------------------------- ciphertext wire format (host parsing, no GPU)
def _wire():
    import importlib.util

    spec = importlib.util.spec_from_file_location("wire_format", os.path.join(ROOT, "oracle", "wire_format.py"))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    return mod


def test_f3_header_known_answers_of_the_reference_tests():
    """native/tests/seal/serialization.cpp:48-130 restated: the header is 16 bytes, a default one is valid, a wrong
    magic / major version / compression mode is not, and a SEAL 3.4 header is upgraded on load."""
    W = _wire()
    assert W.HEADER.size == 16 == W.HEADER_SIZE
    assert W.is_valid_header(W.header(256))
    bad = bytearray(W.header(256))
    bad[0:2] = (0x1212).to_bytes(2, "little")
    assert not W.is_valid_header(bytes(bad))
    assert not W.is_valid_header(W.header(256, version=(2, 5)))
    assert not W.is_valid_header(W.header(256, compr_mode=2))
    h = W.load_header(W.header(256))
    assert (h["magic"], h["header_size"], h["version_major"], h["version_minor"], h["reserved"], h["size"]) == \
        (0xA15E, 0x10, 3, 5, 0, 256)
    up = W.load_header(W.header_3_4(0xF3F3))
    assert up["size"] == 0xF3F3 and up["compr_mode"] == 0 and W.is_valid_header(W.header(up["size"], up["compr_mode"]))
    assert W.load_header(W.header_3_4(0xF3F3), try_upgrade=False)["header_size"] != 0x10  # no upgrade requested


def test_f3_peek_matches_the_oracle_and_reports_errors_like_the_reference():
    import sealhip as S

    W = _wire()
    rng = np.random.default_rng(0)
    n, k, size = 64, 3, 2
    words = rng.integers(0, 2**40, size=size * k * n, dtype=np.uint64)
    pid = (11, 22, 33, 44)
    raw = W.save_ciphertext(pid, True, size, n, k, 2.0**40, words)
    assert len(raw) == 16 + 65 + 16 + 8 + 8 * len(words)  # Ciphertext::save_size, ciphertext.cpp:135-168
    back = W.load_ciphertext(raw)
    assert back["parms_id"] == pid and back["is_ntt_form"] and np.array_equal(back["words"], words)
    info = S.ciphertext_peek(raw)
    assert tuple(info.parms_id) == pid and info.is_ntt_form == 1 and info.size == size
    assert info.coeff_modulus_size == k and info.poly_modulus_degree == n and info.scale == 2.0**40
    assert info.data_words == len(words) and info.total_bytes == len(raw) and info.seeded == 0
    assert S.ciphertext_peek(raw + b"trailing bytes of the next object").total_bytes == len(raw)
    # seeded form: one polynomial + 64 bytes of seed (ciphertext.cpp:189-208)
    seeded = W.save_ciphertext(pid, False, 2, n, k, 1.0, words[: k * n], seed=bytes(range(64)))
    si = S.ciphertext_peek(seeded)
    assert si.seeded == 1 and si.data_words == k * n
    # a SEAL 3.4 outer header is upgraded (serialization.cpp:147-164)
    old = W.header_3_4(len(raw)) + raw[16:]
    assert S.ciphertext_peek(old).total_bytes == len(raw)
    with pytest.raises(S.LogicError, match="loaded SEALHeader is invalid"):
        S.ciphertext_peek(b"\x12\x12" + raw[2:])
    with pytest.raises(S.LogicError, match="incompatible version"):
        S.ciphertext_peek(raw[:3] + b"\x02" + raw[4:])
    with pytest.raises(S.LogicError, match="loaded SEALHeader is invalid"):
        S.ciphertext_peek(raw[:5] + b"\x01" + raw[6:])  # deflate: the reference build has no zlib
    with pytest.raises(RuntimeError, match="I/O error"):
        S.ciphertext_peek(raw[:-8])
    with pytest.raises(RuntimeError, match="I/O error"):
        S.ciphertext_peek(raw[:10])


def test_f3_seed_expansion_blake2xb_known_answers_and_oracle():
    """Ciphertext::expand_seed (ciphertext.cpp:126-133) without the host library: the engine's BLAKE2Xb
    (csrc/blake2xb.cpp, written from the BLAKE2 specifications) and the oracle's (oracle/sealref.c) both reproduce
    tests/golden/prng_vectors.json -- outputs of the REFERENCE's own blake2b.c / blake2xb.c compiled from /root/reference
    (tests/golden/make_prng_vectors.py) -- and the engine's BlakePRNG + sample_poly_uniform equal the oracle's word for
    word, across buffer refills and rejections."""
    import json
    import struct

    import sealhip as S

    vec = json.load(open(os.path.join(ROOT, "tests", "golden", "prng_vectors.json")))
    for impl in (S.blake2xb, O.blake2xb):
        for t in vec["blake2xb"]:
            out = impl(t["outlen"], bytes.fromhex(t["data"]), bytes.fromhex(t["key"]))
            assert len(out) == t["outlen"] and out[:64].hex() == t["head"] and out[-16:].hex() == t["tail"]
        for t in vec["prng"]:
            key = struct.pack("<8Q", *[int(x) for x in t["seed"]])
            b0, b1 = impl(4096, struct.pack("<Q", 0), key), impl(4096, struct.pack("<Q", 1), key)
            assert b0[:32].hex() == t["buffer0_head"] and b0[-32:].hex() == t["buffer0_tail"]
            assert b1[:32].hex() == t["buffer1_head"]
    # 60-bit primes reject about one candidate in eight, small ones almost never: both ends, several buffers deep
    for logn, bits in ((10, [60, 59, 30]), (12, [36, 36, 37]), (13, [55] * 3)):
        n = 1 << logn
        mods = O.coeff_modulus_create(n, bits)
        ctx = S.Context(S.SCHEME_CKKS, logn, mods, 1, 0, device=-1)
        for seed in ([0] * 8, list(range(11, 19)), [2**64 - 1 - i for i in range(8)]):
            for rows in (1, len(mods)):
                got = ctx.expand_seed(rows, seed)
                assert np.array_equal(got, O.expand_seed(seed, mods[:rows], n))
                assert all((got[j] < mods[j]).all() for j in range(rows))
    with pytest.raises(ValueError):
        S.blake2xb(0, b"")
