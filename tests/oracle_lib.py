"""ctypes loader for the CPU oracle (oracle/libsealref.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and bench.py's
cpu_baseline leg. The product (gemini-seal_amd/) never imports this module.
"""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
_LIB = None

u64 = C.c_uint64
u64p = C.POINTER(C.c_uint64)
szt = C.c_size_t


class Modulus(C.Structure):
    _fields_ = [("value", u64), ("cr", u64 * 3), ("bit_count", C.c_int)]


class NttTables(C.Structure):
    _fields_ = [
        ("logn", C.c_int),
        ("n", szt),
        ("mod", Modulus),
        ("root", u64),
        ("inv_degree", u64),
        ("scaled_inv_degree", u64),
        ("reduce_precomp", u64),
        ("root_powers", u64p),
        ("scaled_root_powers", u64p),
        ("inv_root_powers", u64p),
        ("scaled_inv_root_powers", u64p),
    ]


class CkksEncoder(C.Structure):
    _fields_ = [("logn", C.c_int), ("n", szt), ("index_map", C.POINTER(C.c_uint32)), ("roots", C.POINTER(C.c_double)),
                ("inv_roots", C.POINTER(C.c_double))]


class BaseConverter(C.Structure):
    _fields_ = [
        ("isize", szt),
        ("osize", szt),
        ("ibase", C.POINTER(Modulus)),
        ("obase", C.POINTER(Modulus)),
        ("inv_punct", u64p),
        ("matrix", u64p),
    ]


class RnsTool(C.Structure):
    _fields_ = [
        ("n", szt),
        ("logn", C.c_int),
        ("q_size", szt),
        ("B_size", szt),
        ("Bsk_size", szt),
        ("q", C.POINTER(Modulus)),
        ("Bsk", C.POINTER(Modulus)),
        ("m_tilde", Modulus),
        ("m_sk", Modulus),
        ("gamma", Modulus),
        ("t", Modulus),
        ("Bsk_ntt", C.POINTER(NttTables)),
        ("q_to_Bsk", BaseConverter),
        ("q_to_m_tilde", BaseConverter),
        ("B_to_q", BaseConverter),
        ("B_to_m_sk", BaseConverter),
        ("prod_B_mod_q", u64p),
        ("inv_prod_q_mod_Bsk", u64p),
        ("inv_prod_B_mod_m_sk", u64),
        ("inv_m_tilde_mod_Bsk", u64p),
        ("inv_prod_q_mod_m_tilde", u64),
        ("prod_q_mod_Bsk", u64p),
        ("inv_q_last_mod_q", u64p),
    ]


class Context(C.Structure):
    _fields_ = [
        ("scheme", C.c_int),
        ("logn", C.c_int),
        ("n", szt),
        ("n_key", szt),
        ("nsp", szt),
        ("k_first", szt),
        ("t", u64),
        ("mode", C.c_int),
        ("key_mod", C.POINTER(Modulus)),
        ("key_tables", C.POINTER(NttTables)),
        ("rns_tools", C.POINTER(C.POINTER(RnsTool))),
    ]


def build():
    subprocess.check_call(["make", "-s", "-C", ORACLE_DIR])


def lib():
    global _LIB
    if _LIB is not None:
        return _LIB
    path = os.path.join(ORACLE_DIR, "libsealref.so")
    src = os.path.join(ORACLE_DIR, "sealref.c")
    if os.environ.get("SEALREF_LIBRARY"):  # another build of the same source (bench.py: -march=native for its host)
        path = os.environ["SEALREF_LIBRARY"]
    elif not os.path.exists(path) or os.path.getmtime(path) < os.path.getmtime(src):
        build()
    L = C.CDLL(path)
    L.ref_splitmix64.restype = u64
    L.ref_fnv1a64.restype = u64
    L.ref_barrett_reduce_128.restype = u64
    L.ref_barrett_reduce_128.argtypes = [u64, u64, C.POINTER(Modulus)]
    L.ref_barrett_reduce_63.restype = u64
    L.ref_barrett_reduce_63.argtypes = [u64, C.POINTER(Modulus)]
    L.ref_multiply_uint_mod.restype = u64
    L.ref_multiply_uint_mod.argtypes = [u64, u64, C.POINTER(Modulus)]
    L.ref_multiply_add_uint_mod.restype = u64
    L.ref_multiply_add_uint_mod.argtypes = [u64, u64, u64, C.POINTER(Modulus)]
    L.ref_dot_product_mod.restype = u64
    L.ref_dot_product_mod.argtypes = [C.c_void_p, C.c_void_p, szt, C.POINTER(Modulus)]
    L.ref_exponentiate_uint_mod.restype = u64
    L.ref_exponentiate_uint_mod.argtypes = [u64, u64, C.POINTER(Modulus)]
    L.ref_shoupify.restype = u64
    L.ref_shoupify.argtypes = [u64, u64]
    L.ref_modulus_init.argtypes = [C.POINTER(Modulus), u64]
    L.ref_try_invert_uint_mod.argtypes = [u64, u64, u64p]
    L.ref_is_prime.argtypes = [u64]
    L.ref_get_primes.argtypes = [szt, C.c_int, szt, C.c_void_p]
    L.ref_coeff_modulus_create.argtypes = [szt, C.c_void_p, szt, C.c_void_p]
    L.ref_ntt_tables_init.argtypes = [C.POINTER(NttTables), C.c_int, u64]
    L.ref_ntt_tables_free.argtypes = [C.POINTER(NttTables)]
    for f in ("ref_ntt_forward_lazy", "ref_ntt_forward"):
        getattr(L, f).argtypes = [C.c_void_p, C.POINTER(NttTables), C.c_int]
        getattr(L, f).restype = None
    for f in ("ref_ntt_inverse_lazy", "ref_ntt_inverse"):
        getattr(L, f).argtypes = [C.c_void_p, C.POINTER(NttTables)]
        getattr(L, f).restype = None
    L.ref_dyadic_product_coeffmod.argtypes = [C.c_void_p, C.c_void_p, szt, C.POINTER(Modulus), C.c_void_p]
    L.ref_multiply_poly_scalar_coeffmod.argtypes = [C.c_void_p, szt, u64, C.POINTER(Modulus), C.c_void_p]
    L.ref_add_poly_coeffmod.argtypes = [C.c_void_p, C.c_void_p, szt, C.POINTER(Modulus), C.c_void_p]
    L.ref_sub_poly_coeffmod.argtypes = [C.c_void_p, C.c_void_p, szt, C.POINTER(Modulus), C.c_void_p]
    L.ref_negate_poly_coeffmod.argtypes = [C.c_void_p, szt, C.POINTER(Modulus), C.c_void_p]
    L.ref_modulo_poly_coeffs_63.argtypes = [C.c_void_p, szt, C.POINTER(Modulus), C.c_void_p]
    L.ref_base_converter_init.argtypes = [C.POINTER(BaseConverter), C.c_void_p, szt, C.c_void_p, szt]
    L.ref_base_converter_free.argtypes = [C.POINTER(BaseConverter)]
    L.ref_fast_convert.argtypes = [C.POINTER(BaseConverter), C.c_void_p, C.c_void_p]
    L.ref_fast_convert_array.argtypes = [C.POINTER(BaseConverter), C.c_void_p, szt, C.c_void_p]
    L.ref_rns_tool_init.argtypes = [C.POINTER(RnsTool), szt, C.c_void_p, szt, u64]
    L.ref_rns_tool_free.argtypes = [C.POINTER(RnsTool)]
    for f in ("ref_fastbconv_m_tilde", "ref_sm_mrq", "ref_fast_floor", "ref_fastbconv_sk"):
        getattr(L, f).argtypes = [C.POINTER(RnsTool), C.c_void_p, C.c_void_p]
        getattr(L, f).restype = None
    L.ref_divide_and_round_q_last_inplace.argtypes = [C.POINTER(RnsTool), C.c_void_p]
    L.ref_divide_and_round_q_last_ntt_inplace.argtypes = [C.POINTER(RnsTool), C.c_void_p, C.POINTER(NttTables), C.c_int]
    L.ref_galois_elt_from_step.restype = C.c_uint32
    L.ref_galois_elt_from_step.argtypes = [szt, C.c_int, C.POINTER(C.c_int)]
    L.ref_galois_table_ntt.argtypes = [C.c_int, C.c_uint32, C.c_void_p]
    L.ref_apply_galois.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.POINTER(Modulus), C.c_void_p]
    L.ref_apply_galois_ntt.argtypes = [C.c_void_p, C.c_int, C.c_uint32, C.c_void_p]
    L.ref_modup_rns.argtypes = [C.c_void_p, C.c_void_p, szt, szt, szt, szt, C.POINTER(Modulus), szt]
    L.ref_rescale_special_rns_inplace.argtypes = [C.c_void_p, C.c_int, szt, szt, szt, C.POINTER(Modulus), szt,
                                                  C.POINTER(NttTables), C.c_int]
    L.ref_context_init.argtypes = [C.POINTER(Context), C.c_int, C.c_int, C.c_void_p, szt, szt, u64, C.c_int]
    L.ref_context_free.argtypes = [C.POINTER(Context)]
    L.ref_context_rns_tool.restype = C.POINTER(RnsTool)
    L.ref_context_rns_tool.argtypes = [C.POINTER(Context), szt]
    L.ref_bfv_multiply.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p, szt, C.c_void_p]
    L.ref_ckks_multiply.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p, szt, C.c_void_p]
    L.ref_bfv_square.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_ckks_square.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_switch_key_inplace.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_void_p, C.c_void_p]
    L.ref_switch_key_partial.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_void_p, szt, szt, C.c_void_p]
    L.ref_switch_key_finish.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_void_p]
    L.ref_relinearize.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.POINTER(C.c_void_p)]
    L.ref_mod_switch_scale_to_next.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_mod_switch_drop_to_next.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_apply_galois_inplace.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_uint32, C.c_void_p]
    L.ref_evaluator_negate.restype = None
    L.ref_evaluator_negate.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    for fn in (L.ref_evaluator_add, L.ref_evaluator_sub):
        fn.restype = None
        fn.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p, szt, C.c_void_p]
    L.ref_multiply_plain_ntt.restype = None
    L.ref_multiply_plain_ntt.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_multiply_plain.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_void_p]
    L.ref_is_transparent.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt]
    i8p = C.c_void_p
    L.ref_sample_ternary.restype = None
    L.ref_sample_ternary.argtypes = [i8p, szt, u64p]
    L.ref_small_poly_to_rns.restype = None
    L.ref_small_poly_to_rns.argtypes = [C.POINTER(Context), i8p, szt, C.c_int, C.c_void_p]
    L.ref_encrypt_zero_symmetric.restype = None
    L.ref_encrypt_zero_symmetric.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_int, u64p, C.c_void_p]
    L.ref_bfv_encrypt_symmetric.restype = None
    L.ref_bfv_encrypt_symmetric.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_void_p, u64p, C.c_void_p]
    L.ref_generate_kswitch_key.restype = None
    L.ref_generate_kswitch_key.argtypes = [C.POINTER(Context), C.c_void_p, C.c_void_p, u64p, C.c_void_p]
    L.ref_dot_product_ct_sk.restype = None
    L.ref_dot_product_ct_sk.argtypes = [C.POINTER(Context), szt, C.c_void_p, szt, C.c_int, C.c_void_p, C.c_void_p]
    L.ref_decrypt_scale_and_round.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_void_p]
    L.ref_encrypt_zero_symmetric_given.restype = None
    L.ref_encrypt_zero_symmetric_given.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                   C.c_void_p]
    L.ref_encrypt_zero_asymmetric_given.restype = None
    L.ref_encrypt_zero_asymmetric_given.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p,
                                                    C.c_void_p]
    L.ref_multiply_add_plain_with_scaling_variant.restype = None
    L.ref_multiply_add_plain_with_scaling_variant.argtypes = [C.POINTER(Context), szt, C.c_void_p, C.c_int, C.c_void_p]
    L.ref_batch_index_map.restype = None
    L.ref_batch_index_map.argtypes = [C.c_int, C.c_void_p]
    L.ref_batch_encode.restype = None
    L.ref_batch_encode.argtypes = [C.POINTER(NttTables), C.c_void_p, szt, C.c_void_p]
    L.ref_batch_decode.restype = None
    L.ref_batch_decode.argtypes = [C.POINTER(NttTables), C.c_void_p, szt, C.c_void_p]
    for fn in (L.ref_batch_encode_signed, L.ref_batch_decode_signed):
        fn.restype = None
        fn.argtypes = [C.POINTER(NttTables), C.c_void_p, szt, C.c_void_p]
    L.ref_ckks_encoder_init.argtypes = [C.POINTER(CkksEncoder), C.c_int]
    L.ref_ckks_encoder_free.argtypes = [C.POINTER(CkksEncoder)]
    L.ref_ckks_encoder_free.restype = None
    L.ref_ckks_encode.argtypes = [C.POINTER(Context), C.POINTER(CkksEncoder), szt, C.c_void_p, szt, C.c_double, C.c_void_p]
    L.ref_ckks_decode.argtypes = [C.POINTER(Context), C.POINTER(CkksEncoder), szt, C.c_void_p, C.c_double, C.c_void_p]
    L.ref_ckks_encode_value.argtypes = [C.POINTER(Context), szt, C.c_double, C.c_double, C.c_void_p]
    L.ref_ckks_encode_int64.argtypes = [C.POINTER(Context), szt, C.c_int64, C.c_void_p]
    L.ref_blake2xb.argtypes = [C.c_void_p, szt, C.c_void_p, szt, C.c_void_p, szt]
    L.ref_expand_seed.restype = None
    L.ref_expand_seed.argtypes = [C.c_void_p, C.c_void_p, szt, szt, C.c_void_p]
    L.ref_fill_rows.argtypes = [C.c_void_p, szt, szt, C.c_void_p, u64p]
    L.ref_fnv1a64.argtypes = [C.c_void_p, szt]
    L.ref_splitmix64.argtypes = [u64p]
    _LIB = L
    return L


class CkksRef:
    """CKKSEncoder restated (ckks.cpp:14-77, ckks.h:405-747) on top of a RefContext"""

    def __init__(self, ref):
        self.ref, self.enc = ref, CkksEncoder()
        assert lib().ref_ckks_encoder_init(C.byref(self.enc), int(ref.c.logn)) == 0
        self.n = int(ref.c.n)

    def encode(self, values, rows, scale):
        values = np.ascontiguousarray(values, dtype=np.complex128)
        out = np.zeros((rows, self.n), dtype=np.uint64)
        rc = lib().ref_ckks_encode(C.byref(self.ref.c), C.byref(self.enc), rows, values.ctypes.data_as(C.c_void_p),
                                   len(values), scale, ptr(out))
        return rc, out

    def decode(self, plain, scale):
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        out = np.zeros(self.n // 2, dtype=np.complex128)
        rc = lib().ref_ckks_decode(C.byref(self.ref.c), C.byref(self.enc), plain.shape[0], ptr(plain), scale,
                                   out.ctypes.data_as(C.c_void_p))
        assert rc == 0
        return out

    def __del__(self):
        try:
            lib().ref_ckks_encoder_free(C.byref(self.enc))
        except Exception:
            pass


def ptr(a):
    assert a.dtype in (np.uint64, np.int32, np.uint32) and a.flags["C_CONTIGUOUS"]
    return a.ctypes.data_as(C.c_void_p)


def modulus(value):
    m = Modulus()
    rc = lib().ref_modulus_init(C.byref(m), value)
    assert rc == 0, value
    return m


def get_primes(n, bits, count):
    out = np.zeros(count, dtype=np.uint64)
    assert lib().ref_get_primes(n, bits, count, ptr(out)) == 0
    return [int(x) for x in out]


def coeff_modulus_create(n, bit_sizes):
    bs = np.array(bit_sizes, dtype=np.int32)
    out = np.zeros(len(bit_sizes), dtype=np.uint64)
    assert lib().ref_coeff_modulus_create(n, bs.ctypes.data_as(C.c_void_p), len(bit_sizes), ptr(out)) == 0
    return [int(x) for x in out]


class Tables:
    def __init__(self, logn, p):
        self.t = NttTables()
        assert lib().ref_ntt_tables_init(C.byref(self.t), logn, p) == 0, (logn, p)
        self.logn, self.n, self.p = logn, 1 << logn, p

    def arr(self, name):
        return np.ctypeslib.as_array(getattr(self.t, name), shape=(self.n,)).copy()

    def __del__(self):
        try:
            lib().ref_ntt_tables_free(C.byref(self.t))
        except Exception:
            pass


class SplitMix:
    def __init__(self, seed):
        self.state = u64(seed)

    def set(self, seed):
        self.state = u64(seed)

    def fill(self, rows, n, moduli):
        out = np.empty((rows, n), dtype=np.uint64)
        mods = np.array(moduli, dtype=np.uint64)
        assert len(mods) == rows
        lib().ref_fill_rows(ptr(out), rows, n, ptr(mods), C.byref(self.state))
        return out


def blake2xb(outlen, data, key=b""):
    out = C.create_string_buffer(outlen)
    assert lib().ref_blake2xb(out, outlen, data, len(data), key if key else None, len(key)) == 0
    return out.raw


def expand_seed(seed_words, moduli, n):
    """Ciphertext::expand_seed (ciphertext.cpp:126-133): the rows x n words of c_1"""
    seed = np.array([int(s) for s in seed_words], dtype=np.uint64)
    mods = np.array([int(q) for q in moduli], dtype=np.uint64)
    out = np.zeros((len(mods), n), dtype=np.uint64)
    lib().ref_expand_seed(ptr(seed), ptr(mods), len(mods), n, ptr(out))
    return out


def fnv(a):
    a = np.ascontiguousarray(a, dtype=np.uint64)
    return int(lib().ref_fnv1a64(ptr(a), a.size))


class RefContext:
    def __init__(self, scheme, logn, key_moduli, nsp=1, t=0, mode=0):
        self.c = Context()
        km = np.array(key_moduli, dtype=np.uint64)
        rc = lib().ref_context_init(C.byref(self.c), scheme, logn, ptr(km), len(km), nsp, t, mode)
        assert rc == 0
        self.scheme, self.logn, self.n = scheme, logn, 1 << logn
        self.key_moduli = [int(x) for x in key_moduli]
        self.n_key, self.nsp, self.k_first, self.t = len(km), nsp, len(km) - nsp, t

    def rns_tool(self, k):
        r = lib().ref_context_rns_tool(C.byref(self.c), k)
        assert r
        return r

    def tables(self, i):
        return C.byref(self.c.key_tables[i])

    def __del__(self):
        try:
            lib().ref_context_free(C.byref(self.c))
        except Exception:
            pass


class Client:
    """Secret-key side of a BFV/CKKS session built from the oracle's restatements: key generation,
    symmetric encryption, key-switch keys, decryption. Test infrastructure for the semantic end-to-end checks."""

    def __init__(self, ref, seed=1):
        self.ref = ref
        c = ref.c
        self.n, self.n_key, self.k = int(c.n), int(c.n_key), int(c.k_first)
        self.state = C.c_uint64(seed)
        s = np.zeros(self.n, dtype=np.int8)
        lib().ref_sample_ternary(s.ctypes.data, self.n, C.byref(self.state))
        self.s = s
        self.sk = np.zeros((self.n_key, self.n), dtype=np.uint64)  # NTT form, every key prime
        lib().ref_small_poly_to_rns(C.byref(c), s.ctypes.data, self.n_key, 1, ptr(self.sk))
        self.mods = [int(c.key_mod[i].value) for i in range(self.n_key)]

    def sk_powers(self, count):
        """s, s^2, ... in NTT form with key-level row stride (Decryptor::compute_secret_key_array)"""
        out = np.zeros((count, self.n_key, self.n), dtype=np.uint64)
        cur = self.sk.copy()
        for i in range(count):
            out[i] = cur
            nxt = np.zeros_like(cur)
            for r in range(self.n_key):
                lib().ref_dyadic_product_coeffmod(ptr(cur[r]), ptr(self.sk[r]), self.n, C.byref(self.ref.c.key_mod[r]), ptr(nxt[r]))
            cur = nxt
        return out

    def encrypt_bfv(self, plain):
        ct = np.zeros((2, self.k, self.n), dtype=np.uint64)
        plain = np.ascontiguousarray(plain, dtype=np.uint64)
        lib().ref_bfv_encrypt_symmetric(C.byref(self.ref.c), self.k, ptr(self.sk), ptr(plain), C.byref(self.state), ptr(ct))
        return ct

    def encrypt_poly_ntt(self, coeffs):
        """CKKS-style: encrypt an integer polynomial (already scaled), NTT form, at the first ciphertext level"""
        ct = np.zeros((2, self.k, self.n), dtype=np.uint64)
        lib().ref_encrypt_zero_symmetric(C.byref(self.ref.c), self.k, ptr(self.sk), 1, C.byref(self.state), ptr(ct))
        for r in range(self.k):
            p = self.mods[r]
            row = np.array([int(v) % p for v in coeffs], dtype=np.uint64)
            lib().ref_ntt_forward(ptr(row), C.byref(self.ref.c.key_tables[r]), 0)
            lib().ref_add_poly_coeffmod(ptr(ct[0, r]), ptr(row), self.n, C.byref(self.ref.c.key_mod[r]), ptr(ct[0, r]))
        return ct

    def centered_from_ntt_rows(self, rows):
        """k NTT-form rows -> centred integer coefficients (CRT over the first k key primes)"""
        k = rows.shape[0]
        res = []
        for r in range(k):
            row = np.ascontiguousarray(rows[r]).copy()
            lib().ref_ntt_inverse(ptr(row), C.byref(self.ref.c.key_tables[r]))
            res.append([int(v) for v in row])
        q = 1
        for r in range(k):
            q *= self.mods[r]
        out = [0] * self.n
        for r in range(k):
            qr = q // self.mods[r]
            inv = pow(qr % self.mods[r], -1, self.mods[r])
            for i in range(self.n):
                out[i] = (out[i] + res[r][i] * inv % self.mods[r] * qr) % q
        return [v - q if v > q // 2 else v for v in out], q

    def kswitch_key(self, new_key_ntt):
        d = (self.k + int(self.ref.c.nsp) - 1) // int(self.ref.c.nsp)
        key = np.zeros((d, 2, self.n_key, self.n), dtype=np.uint64)
        new_key_ntt = np.ascontiguousarray(new_key_ntt[: self.k], dtype=np.uint64)
        lib().ref_generate_kswitch_key(C.byref(self.ref.c), ptr(self.sk), ptr(new_key_ntt), C.byref(self.state), ptr(key))
        return key

    def relin_key(self):
        return self.kswitch_key(self.sk_powers(2)[1])

    def galois_key(self, elt):
        """key for s(x^elt) (KeyGenerator::galois_keys: apply_galois_ntt on the secret key)"""
        rot = np.zeros((self.k, self.n), dtype=np.uint64)
        for r in range(self.k):
            lib().ref_apply_galois_ntt(ptr(self.sk[r]), int(self.ref.c.logn), elt, ptr(rot[r]))
        return self.kswitch_key(rot)

    def decrypt_bfv(self, ct, k=None):
        k = k or ct.shape[1]
        size = ct.shape[0]
        ct = np.ascontiguousarray(ct, dtype=np.uint64)
        dot = np.zeros((k, self.n), dtype=np.uint64)
        pw = self.sk_powers(size - 1)
        lib().ref_dot_product_ct_sk(C.byref(self.ref.c), k, ptr(ct), size, 0, ptr(pw), ptr(dot))
        out = np.zeros(self.n, dtype=np.uint64)
        assert lib().ref_decrypt_scale_and_round(C.byref(self.ref.c), k, ptr(dot), ptr(out)) == 0
        return out


def negacyclic_mod_t(a, b, t):
    """a*b in Z_t[x]/(x^N+1) with exact integer convolution"""
    n = len(a)
    full = np.convolve(np.asarray(a, dtype=object), np.asarray(b, dtype=object))
    res = [int(v) for v in full[:n]]
    for i in range(n, 2 * n - 1):
        res[i - n] -= int(full[i])
    return np.array([v % t for v in res], dtype=np.uint64)
