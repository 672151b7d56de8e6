"""Host-side Python mirror of the reference's Evaluator interface over the sealhip C ABI.

The product path is `libsealhip.so` (hand-written HIP for gfx950 behind include/sealhip.h); this module
only binds it with ctypes and mirrors the names, argument meaning and error behaviour of
seal::Evaluator (native/src/seal/evaluator.h) so that tests read like the reference's own.
There is NO CPU fallback: if the library or a HIP device is missing, calls raise.

Error mapping (native/src/seal/c/defines.h:34-49 <-> evaluator.cpp exceptions):
  E_INVALIDARG -> ValueError (std::invalid_argument), COR_E_INVALIDOPERATION -> LogicError
  (std::logic_error), E_POINTER -> TypeError, E_OUTOFMEMORY -> MemoryError, E_UNEXPECTED -> RuntimeError.
"""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# SEALHIP_LIBRARY selects another build of the same HIP library (tools/: the measurement-only build); there is no
# non-HIP implementation to select.
LIB_PATH = os.environ.get("SEALHIP_LIBRARY") or os.path.join(os.path.dirname(_HERE), "lib", "libsealhip.so")

S_OK = 0
E_POINTER = 0x80004003
E_INVALIDARG = 0x80070057
E_OUTOFMEMORY = 0x8007000E
E_UNEXPECTED = 0x8000FFFF
COR_E_INVALIDOPERATION = 0x80131509

SCHEME_BFV, SCHEME_CKKS = 1, 2
MODE_PARITY, MODE_STRICT = 0, 1
BASE_Q, BASE_BSK, BASE_KEY = 0, 1, 2


class LogicError(RuntimeError):
    """std::logic_error of the reference (unsupported operation for the scheme, host-only context, ...)."""


class Params(C.Structure):
    _fields_ = [
        ("scheme", C.c_uint32),
        ("log_n", C.c_uint32),
        ("n_key_moduli", C.c_uint32),
        ("n_special_primes", C.c_uint32),
        ("key_moduli", C.POINTER(C.c_uint64)),
        ("plain_modulus", C.c_uint64),
        ("mode", C.c_uint32),
        ("device", C.c_int32),
    ]


_lib = None

# every symbol include/sealhip.h declares: name -> argtypes (restype is always long unless noted)
_vp, _u32, _u64, _sz, _i32 = C.c_void_p, C.c_uint32, C.c_uint64, C.c_size_t, C.c_int32
SYMBOLS = {
    "sealhip_last_error_string": None,
    "sealhip_num_devices": [C.POINTER(C.c_int32)],
    "sealhip_context_create": [C.POINTER(Params), C.POINTER(_vp)],
    "sealhip_context_destroy": [_vp],
    "sealhip_context_first_level": [_vp, C.POINTER(_u32)],
    "sealhip_context_bsk_size": [_vp, _u32, C.POINTER(_u32)],
    "sealhip_set_stream": [_vp, _vp],
    "sealhip_use_default_stream": [_vp],
    "sealhip_synchronize": [_vp],
    "sealhip_context_lane_count": [_vp, C.POINTER(_u32)],
    "sealhip_debug_ntt_handoff": [_vp, _u32, _i32],
    "sealhip_malloc": [_vp, _sz, C.POINTER(_vp)],
    "sealhip_free": [_vp, _vp],
    "sealhip_memcpy_h2d": [_vp, _vp, _vp, _sz],
    "sealhip_memcpy_d2h": [_vp, _vp, _vp, _sz],
    "sealhip_profile_enable": [_vp, _i32],
    "sealhip_profile_fetch": [_vp, C.c_char_p, _sz],
    "sealhip_debug_ntt_table": [_vp, _u32, _u32, _vp, _sz],
    "sealhip_debug_rns_constants": [_vp, _u32, _u32, _vp, _sz, C.POINTER(_sz)],
    "sealhip_ntt_negacyclic_harvey_lazy": [_vp, _vp, _sz, _u32, _u32],
    "sealhip_ntt_negacyclic_harvey": [_vp, _vp, _sz, _u32, _u32],
    "sealhip_inverse_ntt_negacyclic_harvey_lazy": [_vp, _vp, _sz, _u32, _u32],
    "sealhip_inverse_ntt_negacyclic_harvey": [_vp, _vp, _sz, _u32, _u32],
    "sealhip_dyadic_product_coeffmod": [_vp, _vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_multiply_poly_scalar_coeffmod": [_vp, _vp, _sz, _u32, _u32, _u64, _vp],
    "sealhip_add_poly_coeffmod": [_vp, _vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_sub_poly_coeffmod": [_vp, _vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_negate_poly_coeffmod": [_vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_fastbconv_m_tilde": [_vp, _u32, _vp, _sz, _vp],
    "sealhip_sm_mrq": [_vp, _u32, _vp, _sz, _vp],
    "sealhip_fast_floor": [_vp, _u32, _vp, _sz, _vp],
    "sealhip_fastbconv_sk": [_vp, _u32, _vp, _sz, _vp],
    "sealhip_divide_and_round_q_last_inplace": [_vp, _u32, _vp, _sz],
    "sealhip_divide_and_round_q_last_ntt_inplace": [_vp, _u32, _vp, _sz],
    "sealhip_galois_elt_from_step": [_vp, _i32, C.POINTER(_u32)],
    "sealhip_apply_galois": [_vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_apply_galois_ntt": [_vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_kswitch_key_load": [_vp, _vp, _u32, _i32, C.POINTER(_vp)],
    "sealhip_kswitch_key_destroy": [_vp, _vp],
    "sealhip_modup_rns": [_vp, _u32, _u32, _vp, _sz],
    "sealhip_rescale_special_rns_inplace": [_vp, _u32, _vp, _sz],
    "sealhip_switch_key_inplace": [_vp, _u32, _vp, _vp, _sz, _vp],
    "sealhip_switch_key_partial": [_vp, _u32, _vp, _sz, _vp, _u32, _u32, _vp],
    "sealhip_switch_key_finish": [_vp, _u32, _vp, _vp, _sz, _u32],
    "sealhip_kswitch_digits": [_vp, _u32, _vp],
    "sealhip_evaluator_multiply": [_vp, _u32, _vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_square": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_relinearize": [_vp, _u32, _vp, _u32, _sz, C.POINTER(_vp), _u32],
    "sealhip_evaluator_multiply_many": [_vp, _u32, _vp, _u32, _sz, C.POINTER(_vp), _u32, _vp],
    "sealhip_evaluator_exponentiate": [_vp, _u32, _vp, _u64, _sz, C.POINTER(_vp), _u32, _vp],
    "sealhip_evaluator_multiply_host": [_vp, _u32, _vp, _u32, _vp, _u32, _sz, _vp, C.POINTER(_vp), _u32],
    "sealhip_evaluator_relinearize_host": [_vp, _u32, _vp, _u32, _sz, C.POINTER(_vp), _u32],
    "sealhip_evaluator_rotate_vector_host": [_vp, _u32, _vp, _sz, _i32, C.POINTER(_u32), C.POINTER(_vp), _u32],
    "sealhip_evaluator_mod_switch_to_next_host": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_rescale_to_next_host": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_mod_switch_to_next": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_rescale_to_next": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_mod_switch_to_next_strided": [_vp, _u32, _vp, _u32, _sz, _sz, _vp],
    "sealhip_evaluator_rescale_to_next_strided": [_vp, _u32, _vp, _u32, _sz, _sz, _vp],
    "sealhip_evaluator_apply_galois": [_vp, _u32, _vp, _sz, _u32, _vp],
    "sealhip_evaluator_transform_to_ntt": [_vp, _u32, _vp, _u32, _sz],
    "sealhip_evaluator_transform_from_ntt": [_vp, _u32, _vp, _u32, _sz],
    "sealhip_evaluator_negate": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_add": [_vp, _u32, _vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_sub": [_vp, _u32, _vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_evaluator_multiply_plain_ntt": [_vp, _u32, _vp, _u32, _sz, _vp, _sz],
    "sealhip_evaluator_multiply_plain": [_vp, _u32, _vp, _u32, _sz, _vp, _sz],
    "sealhip_is_transparent": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_transparency_sink": [_vp, _vp, _sz],
    "sealhip_debug_butterfly_rate": [_vp, _u32, _u32, _vp],
    "sealhip_debug_chunk_log": [_vp, _vp, _sz, _vp],
    "sealhip_modulo_poly_coeffs_63": [_vp, _vp, _sz, _u32, _u32, _vp],
    "sealhip_evaluator_rotate_vector": [_vp, _u32, _vp, _sz, _i32, C.POINTER(_u32), C.POINTER(_vp), _u32],
    "sealhip_decryptor_dot_product_ct_sk": [_vp, _u32, _vp, _u32, _sz, _vp, _i32, _vp],
    "sealhip_decrypt_scale_and_round": [_vp, _u32, _vp, _sz, _vp],
    "sealhip_encrypt_zero_symmetric": [_vp, _u32, _i32, _vp, _vp, _vp, _sz, _vp],
    "sealhip_encrypt_zero_asymmetric": [_vp, _u32, _i32, _vp, _vp, _vp, _sz, _vp],
    "sealhip_multiply_add_plain_with_scaling_variant": [_vp, _u32, _vp, _sz, _vp, _u32, _sz, _i32],
    "sealhip_evaluator_add_plain": [_vp, _u32, _vp, _u32, _sz, _vp, _sz, _i32],
    "sealhip_context_using_batching": [_vp, C.POINTER(_i32)],
    "sealhip_batch_encode": [_vp, _vp, _sz, _sz, _vp],
    "sealhip_batch_decode": [_vp, _vp, _sz, _vp],
    "sealhip_batch_encode_int64": [_vp, _vp, _sz, _sz, _vp],
    "sealhip_batch_decode_int64": [_vp, _vp, _sz, _vp],
    "sealhip_context_set_parms_id": [_vp, _u32, _vp],
    "sealhip_ciphertext_peek": [_vp, _sz, _vp],
    "sealhip_ciphertext_load": [_vp, _vp, _sz, _vp, _vp, _sz],
    "sealhip_ciphertext_save_size": [_vp, _u32, _u32, C.POINTER(_sz)],
    "sealhip_ciphertext_save": [_vp, _vp, _vp, _vp, _sz, C.POINTER(_sz)],
    "sealhip_is_data_valid_for": [_vp, _u32, _vp, _u32, _sz, _vp],
    "sealhip_ciphertext_resize": [_vp, _u32, _vp, _u32, _vp, _u32, _sz],
    "sealhip_kswitch_key_load_stream": [_vp, _vp, _sz, _u32, C.POINTER(_vp), C.POINTER(_u64)],
    "sealhip_expand_seed_host": [_vp, _u32, _vp, _vp],
    "sealhip_host_register": [_vp, _vp, _sz],
    "sealhip_host_unregister": [_vp, _vp],
    "sealhip_debug_blake2xb": [_vp, _sz, _vp, _sz, _vp, _sz],
    "sealhip_kswitch_keys_save": [_vp, C.POINTER(_vp), _u32, _vp, _sz, C.POINTER(_sz)],
    "sealhip_graph_capture_begin": [_vp],
    "sealhip_graph_capture_end": [_vp, C.POINTER(_vp)],
    "sealhip_graph_launch": [_vp, _vp],
    "sealhip_graph_destroy": [_vp, _vp],
    "sealhip_ckks_encode": [_vp, _u32, _vp, _sz, _sz, C.c_double, _vp],
    "sealhip_ckks_decode": [_vp, _u32, _vp, _sz, C.c_double, _vp],
    "sealhip_ckks_encode_value": [_vp, _u32, C.c_double, C.c_double, _sz, _vp],
    "sealhip_ckks_encode_int64": [_vp, _u32, C.c_int64, _sz, _vp],
}


class CiphertextInfo(C.Structure):
    """sealhip_ciphertext_info: what Ciphertext::save_members writes ahead of the words (ciphertext.cpp:170-188)"""
    _fields_ = [("parms_id", C.c_uint64 * 4), ("is_ntt_form", C.c_uint32), ("size", C.c_uint32),
                ("coeff_modulus_size", C.c_uint32), ("seeded", C.c_uint32), ("poly_modulus_degree", C.c_uint64),
                ("scale", C.c_double), ("data_words", C.c_uint64), ("total_bytes", C.c_uint64)]


def ciphertext_peek(raw):
    """Header + metadata of a serialized ciphertext (no context needed)"""
    info = CiphertextInfo()
    buf = (C.c_char * len(raw)).from_buffer_copy(raw)
    _check(lib().sealhip_ciphertext_peek(C.addressof(buf), len(raw), C.addressof(info)))
    return info


def lib():
    """Load libsealhip.so; fails loudly when the HIP extension has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise ImportError(
                "libsealhip.so is missing (%s): build it with __graft_entry__.build() / make -C gemini-seal_amd; "
                "there is no CPU fallback" % LIB_PATH)
        L = C.CDLL(LIB_PATH)
        for name, argtypes in SYMBOLS.items():
            fn = getattr(L, name)
            if argtypes is None:
                fn.restype = C.c_char_p
                fn.argtypes = []
            else:
                fn.restype = C.c_long
                fn.argtypes = argtypes
        _lib = L
    return _lib


def _check(hr):
    hr &= 0xFFFFFFFF
    if hr == S_OK:
        return
    msg = lib().sealhip_last_error_string().decode("utf-8", "replace")
    if hr == E_INVALIDARG:
        raise ValueError(msg)
    if hr == COR_E_INVALIDOPERATION:
        raise LogicError(msg)
    if hr == E_POINTER:
        raise TypeError(msg)
    if hr == E_OUTOFMEMORY:
        raise MemoryError(msg)
    raise RuntimeError("sealhip error 0x%08x: %s" % (hr, msg))


def num_devices():
    n = C.c_int32(0)
    _check(lib().sealhip_num_devices(C.byref(n)))
    return n.value


def _ptr(x):
    """Device pointer of a DeviceBuffer, a torch CUDA tensor, or a raw integer address."""
    if isinstance(x, DeviceBuffer):
        return x.ptr
    if hasattr(x, "data_ptr"):
        return x.data_ptr()
    return int(x)


class DeviceBuffer:
    """uint64 words in HBM, owned through sealhip_malloc / sealhip_free."""

    def __init__(self, ctx, words):
        self.ctx, self.words = ctx, int(words)
        p = C.c_void_p()
        _check(lib().sealhip_malloc(ctx.handle, self.words * 8, C.byref(p)))
        self.ptr = p.value

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        assert host.size == self.words, (host.size, self.words)
        _check(lib().sealhip_memcpy_h2d(self.ctx.handle, self.ptr, host.ctypes.data, host.size * 8))
        return self

    def download(self, shape=None):
        self.ctx.synchronize()  # also surfaces asynchronous kernel-side failures
        out = np.empty(self.words, dtype=np.uint64)
        _check(lib().sealhip_memcpy_d2h(self.ctx.handle, out.ctypes.data, self.ptr, self.words * 8))
        return out.reshape(shape) if shape is not None else out

    def free(self):
        if self.ptr:
            _check(lib().sealhip_free(self.ctx.handle, self.ptr))
            self.ptr = None

    def __del__(self):
        try:
            self.free()
        except Exception:
            pass


class KSwitchKeys:
    """One key-switch key (n_digits x 2 x n_key x N, keygenerator.cpp:325-369) resident in HBM."""

    def __init__(self, ctx, key, n_digits=None, from_host=True):
        self.ctx = ctx
        if from_host:
            key = np.ascontiguousarray(key, dtype=np.uint64)
            if n_digits is None:
                n_digits = key.shape[0]
            src = key.ctypes.data
        else:
            src = _ptr(key)
        h = C.c_void_p()
        _check(lib().sealhip_kswitch_key_load(ctx.handle, src, n_digits, 1 if from_host else 0, C.byref(h)))
        self.handle = h.value

    @classmethod
    def from_stream(cls, ctx, raw, index):
        """KSwitchKeys::load (kswitchkeys.cpp:87-150): keys_[index] of a serialized RelinKeys / GaloisKeys object, its digits
        copied straight from the byte stream into HBM. Returns None when the slot is empty."""
        buf = (C.c_char * len(raw)).from_buffer_copy(raw)
        h, slots = C.c_void_p(), _u64(0)
        _check(lib().sealhip_kswitch_key_load_stream(ctx.handle, C.addressof(buf), len(raw), index, C.byref(h), C.byref(slots)))
        if not h.value:
            return None
        self = cls.__new__(cls)
        self.ctx, self.handle, self.n_slots = ctx, h.value, slots.value
        return self

    def __del__(self):
        try:
            if self.handle:
                lib().sealhip_kswitch_key_destroy(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


def blake2xb(outlen, data, key=b""):
    """BLAKE2Xb of csrc/blake2xb.cpp (the PRNG under Ciphertext::expand_seed)"""
    out = C.create_string_buffer(outlen)
    _check(lib().sealhip_debug_blake2xb(out, outlen, data, len(data), key if key else None, len(key)))
    return out.raw


def save_kswitch_keys(ctx, keys):
    """KSwitchKeys::save (kswitchkeys.cpp:43-85): keys = list of KSwitchKeys or None (unused slot) -> the reference's byte
    stream, digit words copied straight from HBM"""
    arr = (C.c_void_p * max(1, len(keys)))(*[(k.handle if k is not None else None) for k in keys])
    need = _sz(0)
    _check(lib().sealhip_kswitch_keys_save(ctx.handle, arr, len(keys), None, 0, C.byref(need)))
    buf = (C.c_char * need.value)()
    written = _sz(0)
    _check(lib().sealhip_kswitch_keys_save(ctx.handle, arr, len(keys), C.addressof(buf), need.value, C.byref(written)))
    return bytes(buf[: written.value])


class Graph:
    """An instantiated hipGraph of engine launches (sealhip_graph)."""

    def __init__(self, ctx, handle):
        self.ctx, self.handle = ctx, handle

    def launch(self):
        _check(lib().sealhip_graph_launch(self.ctx.handle, self.handle))

    def __del__(self):
        try:
            if self.handle:
                lib().sealhip_graph_destroy(self.ctx.handle, self.handle)
                self.handle = None
        except Exception:
            pass


class Context:
    """Plain mirror of what the path needs from SEALContext (context.cpp:455-540)."""

    def __init__(self, scheme, log_n, key_moduli, n_special_primes=1, plain_modulus=0, mode=MODE_PARITY, device=0):
        self.scheme, self.log_n, self.n = scheme, log_n, 1 << log_n
        self.key_moduli = [int(x) for x in key_moduli]
        self.n_key, self.nsp = len(self.key_moduli), n_special_primes
        arr = (C.c_uint64 * self.n_key)(*self.key_moduli)
        p = Params(scheme, log_n, self.n_key, n_special_primes, arr, plain_modulus, mode, device)
        h = C.c_void_p()
        _check(lib().sealhip_context_create(C.byref(p), C.byref(h)))
        self.handle = h.value
        self.k_first = self.n_key - n_special_primes

    # -- memory
    def alloc(self, words):
        return DeviceBuffer(self, words)

    def upload(self, host):
        host = np.ascontiguousarray(host, dtype=np.uint64)
        return DeviceBuffer(self, host.size).upload(host)

    def upload_i32(self, host):
        """small signed samples (int32) -> device; returns a DeviceBuffer holding the raw words"""
        host = np.ascontiguousarray(host, dtype=np.int32)
        assert host.size % 2 == 0
        return DeviceBuffer(self, host.size // 2).upload(host.view(np.uint64))

    def synchronize(self):
        _check(lib().sealhip_synchronize(self.handle))

    def set_stream(self, stream_ptr):
        """hipStream_t of the calling thread's lane (None/0: back to a private stream)"""
        _check(lib().sealhip_set_stream(self.handle, stream_ptr))

    def lane_count(self):
        v = C.c_uint32()
        _check(lib().sealhip_context_lane_count(self.handle, C.byref(v)))
        return v.value

    def use_default_stream(self):
        _check(lib().sealhip_use_default_stream(self.handle))

    def debug_ntt_handoff(self, spin_limit=0, suppress_signal=False):
        _check(lib().sealhip_debug_ntt_handoff(self.handle, spin_limit, 1 if suppress_signal else 0))

    def bsk_size(self, k):
        v = C.c_uint32()
        _check(lib().sealhip_context_bsk_size(self.handle, k, C.byref(v)))
        return v.value

    def rows(self, k, base):
        if base == BASE_Q:
            return k
        if base == BASE_BSK:
            return self.bsk_size(k)
        return k + self.nsp

    def profile_enable(self, on=True):
        _check(lib().sealhip_profile_enable(self.handle, 1 if on else 0))

    def profile_fetch(self):
        """{kernel tag: {"launches", "ms", "units"}} from HIP events on the launch stream; clears the records."""
        import json

        buf = C.create_string_buffer(1 << 16)
        _check(lib().sealhip_profile_fetch(self.handle, buf, len(buf)))
        return json.loads(buf.value.decode())

    # -- introspection (host only)
    def debug_ntt_table(self, prime_index, kind):
        out = np.empty(self.n, dtype=np.uint64)
        _check(lib().sealhip_debug_ntt_table(self.handle, prime_index, kind, out.ctypes.data, out.size))
        return out

    def debug_rns_constants(self, k, which):
        out = np.empty(70 * 70, dtype=np.uint64)
        w = C.c_size_t()
        _check(lib().sealhip_debug_rns_constants(self.handle, k, which, out.ctypes.data, out.size, C.byref(w)))
        return out[: w.value].copy()

    def galois_elt_from_step(self, step):
        v = C.c_uint32()
        _check(lib().sealhip_galois_elt_from_step(self.handle, step, C.byref(v)))
        return v.value

    # -- L2 functions (names of native/src/seal/util/*.h)
    def ntt_negacyclic_harvey_lazy(self, data, count, k, base=BASE_Q):
        _check(lib().sealhip_ntt_negacyclic_harvey_lazy(self.handle, _ptr(data), count, k, base))

    def ntt_negacyclic_harvey(self, data, count, k, base=BASE_Q):
        _check(lib().sealhip_ntt_negacyclic_harvey(self.handle, _ptr(data), count, k, base))

    def inverse_ntt_negacyclic_harvey_lazy(self, data, count, k, base=BASE_Q):
        _check(lib().sealhip_inverse_ntt_negacyclic_harvey_lazy(self.handle, _ptr(data), count, k, base))

    def inverse_ntt_negacyclic_harvey(self, data, count, k, base=BASE_Q):
        _check(lib().sealhip_inverse_ntt_negacyclic_harvey(self.handle, _ptr(data), count, k, base))

    def dyadic_product_coeffmod(self, a, b, count, k, result, base=BASE_Q):
        _check(lib().sealhip_dyadic_product_coeffmod(self.handle, _ptr(a), _ptr(b), count, k, base, _ptr(result)))

    def multiply_poly_scalar_coeffmod(self, a, count, k, scalar, result, base=BASE_Q):
        _check(lib().sealhip_multiply_poly_scalar_coeffmod(self.handle, _ptr(a), count, k, base, scalar, _ptr(result)))

    def add_poly_coeffmod(self, a, b, count, k, result, base=BASE_Q):
        _check(lib().sealhip_add_poly_coeffmod(self.handle, _ptr(a), _ptr(b), count, k, base, _ptr(result)))

    def sub_poly_coeffmod(self, a, b, count, k, result, base=BASE_Q):
        _check(lib().sealhip_sub_poly_coeffmod(self.handle, _ptr(a), _ptr(b), count, k, base, _ptr(result)))

    def modulo_poly_coeffs_63(self, a, count, k, result, base=BASE_Q):
        _check(lib().sealhip_modulo_poly_coeffs_63(self.handle, _ptr(a), count, k, base, _ptr(result)))

    def is_transparent(self, ct, size, k, count):
        """Ciphertext::is_transparent (ciphertext.h:471-476) for each ciphertext of the batch -> numpy bool array"""
        flags = np.zeros(count, dtype=np.uint8)
        _check(lib().sealhip_is_transparent(self.handle, k, _ptr(ct), size, count, flags.ctypes.data))
        return flags.astype(bool)

    def transparency_sink(self, flags, capacity):
        """The is_transparent test as a flag output of the Evaluator operations: `flags` = device buffer of `capacity`
        uint32 words (None removes the sink); after an operation flags[i] != 0 iff polynomials 1.. of result i are non-zero."""
        _check(lib().sealhip_transparency_sink(self.handle, _ptr(flags) if flags is not None else None, capacity))

    def dot_product_ct_sk(self, ct, size, k, count, sk_powers_ntt, is_ntt_form, out):
        """Decryptor::dot_product_ct_sk_array (decryptor.cpp:218-265)"""
        _check(lib().sealhip_decryptor_dot_product_ct_sk(self.handle, k, _ptr(ct), size, count, _ptr(sk_powers_ntt),
                                                         1 if is_ntt_form else 0, _ptr(out)))

    def decrypt_scale_and_round(self, k, poly, count, out):
        """RNSTool::decrypt_scale_and_round (rns.cpp:1070-1126)"""
        _check(lib().sealhip_decrypt_scale_and_round(self.handle, k, _ptr(poly), count, _ptr(out)))

    # ---- encrypt-side arithmetic (the random samples come from the caller)
    def encrypt_zero_symmetric(self, rows, is_ntt_form, a_ntt, noise, sk_ntt, count, ct):
        """util::encrypt_zero_symmetric (util/rlwe.cpp:204-300); noise = int32 device words (count x N)"""
        _check(lib().sealhip_encrypt_zero_symmetric(self.handle, rows, 1 if is_ntt_form else 0, _ptr(a_ntt), _ptr(noise),
                                                    _ptr(sk_ntt), count, _ptr(ct)))

    def encrypt_zero_asymmetric(self, rows, is_ntt_form, pk_ntt, u, noise, count, ct):
        """util::encrypt_zero_asymmetric (util/rlwe.cpp:140-202); u = count x N, noise = count x 2 x N int32"""
        _check(lib().sealhip_encrypt_zero_asymmetric(self.handle, rows, 1 if is_ntt_form else 0, _ptr(pk_ntt), _ptr(u),
                                                     _ptr(noise), count, _ptr(ct)))

    def multiply_add_plain_with_scaling_variant(self, k, plain, ct, size, count, plain_stride=None, subtract=False):
        """util/scalingvariant.cpp:15-92 on c_0 of every ciphertext"""
        stride = self.n if plain_stride is None else plain_stride
        _check(lib().sealhip_multiply_add_plain_with_scaling_variant(self.handle, k, _ptr(plain), stride, _ptr(ct), size,
                                                                     count, 1 if subtract else 0))

    def ckks_encode(self, values, k, scale, plain=None):
        """CKKSEncoder::encode (ckks.h:405-617): values = numpy complex array [count][n_values] -> device plaintexts
        [count][k][N] (NTT form)"""
        values = np.ascontiguousarray(values, dtype=np.complex128)
        count, nv = values.shape
        dv = DeviceBuffer(self, values.size * 2).upload(values.view(np.uint64))
        if plain is None:
            plain = self.alloc(count * k * self.n)
        _check(lib().sealhip_ckks_encode(self.handle, k, _ptr(dv), nv, count, float(scale), _ptr(plain)))
        self.synchronize()
        dv.free()
        return plain

    def ckks_encode_value(self, value, k, scale, count=1, plain=None):
        """CKKSEncoder::encode(double value, ...) (ckks.cpp:80-216); an int value takes the int64 overload (:218-275)"""
        if plain is None:
            plain = self.alloc(count * k * self.n)
        if isinstance(value, (int, np.integer)):
            _check(lib().sealhip_ckks_encode_int64(self.handle, k, int(value), count, _ptr(plain)))
        else:
            _check(lib().sealhip_ckks_encode_value(self.handle, k, float(value), float(scale), count, _ptr(plain)))
        return plain

    def ckks_decode(self, plain, k, count, scale):
        """CKKSEncoder::decode (ckks.h:623-747) -> numpy complex array [count][N/2]"""
        out = self.alloc(count * self.n)
        _check(lib().sealhip_ckks_decode(self.handle, k, _ptr(plain), count, float(scale), _ptr(out)))
        res = out.download().view(np.complex128).reshape(count, self.n // 2)
        out.free()
        return res

    # ---- HIP graphs: capture a fixed sequence of operations on fixed buffers, replay it as one launch
    def capture(self, fn):
        """Runs fn() once eagerly (allocations, tables), then captures a second run from the context's stream."""
        fn()
        self.synchronize()
        _check(lib().sealhip_graph_capture_begin(self.handle))
        try:
            fn()
        finally:
            h = C.c_void_p()
            hr = lib().sealhip_graph_capture_end(self.handle, C.byref(h))
        _check(hr)
        return Graph(self, h.value)

    # ---- ciphertext wire format
    def set_parms_id(self, k, parms_id):
        arr = (C.c_uint64 * 4)(*[int(x) for x in parms_id])
        _check(lib().sealhip_context_set_parms_id(self.handle, k, C.addressof(arr)))

    def expand_seed(self, rows, seed_words):
        """Ciphertext::expand_seed (ciphertext.cpp:126-133) on the host: rows x N words of c_1"""
        seed = np.array([int(x) for x in seed_words], dtype=np.uint64)
        out = np.zeros((rows, self.n), dtype=np.uint64)
        _check(lib().sealhip_expand_seed_host(self.handle, rows, seed.ctypes.data, out.ctypes.data))
        return out

    def load_ciphertext(self, raw, dst, capacity_words=None):
        """Ciphertext::load (ciphertext.cpp:228-330): words go from `raw` (host bytes) straight into dst (device)"""
        info = CiphertextInfo()
        buf = (C.c_char * len(raw)).from_buffer_copy(raw)
        cap = dst.words if capacity_words is None else capacity_words
        _check(lib().sealhip_ciphertext_load(self.handle, C.addressof(buf), len(raw), C.addressof(info), _ptr(dst), cap))
        return info

    def save_ciphertext(self, info, src):
        """Ciphertext::save (ciphertext.cpp:170-226), uncompressed -> bytes"""
        need = _sz(0)
        _check(lib().sealhip_ciphertext_save_size(self.handle, info.size, info.coeff_modulus_size, C.byref(need)))
        buf = (C.c_char * need.value)()
        written = _sz(0)
        _check(lib().sealhip_ciphertext_save(self.handle, C.addressof(info), _ptr(src), C.addressof(buf), need.value,
                                             C.byref(written)))
        return bytes(buf[: written.value])

    def is_data_valid_for(self, ct, size, k, count):
        """is_data_valid_for (valcheck.cpp:284-317) per ciphertext of the batch -> numpy bool array"""
        flags = np.zeros(count, dtype=np.uint8)
        _check(lib().sealhip_is_data_valid_for(self.handle, k, _ptr(ct), size, count, flags.ctypes.data))
        return flags.astype(bool)

    # ---- BatchEncoder
    @property
    def using_batching(self):
        flag = _i32(0)
        _check(lib().sealhip_context_using_batching(self.handle, C.byref(flag)))
        return bool(flag.value)

    def batch_encode(self, values, n_values, count, plain):
        """BatchEncoder::encode (batchencoder.cpp:113-154)"""
        _check(lib().sealhip_batch_encode(self.handle, _ptr(values), n_values, count, _ptr(plain)))

    def batch_decode(self, plain, count, values):
        """BatchEncoder::decode (batchencoder.cpp:339-376)"""
        _check(lib().sealhip_batch_decode(self.handle, _ptr(plain), count, _ptr(values)))

    def batch_encode_int64(self, values, n_values, count, plain):
        """BatchEncoder::encode(vector<int64_t>) (batchencoder.cpp:156-198); values = device words holding int64"""
        _check(lib().sealhip_batch_encode_int64(self.handle, _ptr(values), n_values, count, _ptr(plain)))

    def batch_decode_int64(self, plain, count, values):
        """BatchEncoder::decode(vector<int64_t>) (batchencoder.cpp:378-420)"""
        _check(lib().sealhip_batch_decode_int64(self.handle, _ptr(plain), count, _ptr(values)))

    def negate_poly_coeffmod(self, a, count, k, result, base=BASE_Q):
        _check(lib().sealhip_negate_poly_coeffmod(self.handle, _ptr(a), count, k, base, _ptr(result)))

    def fastbconv_m_tilde(self, k, inp, count, out):
        _check(lib().sealhip_fastbconv_m_tilde(self.handle, k, _ptr(inp), count, _ptr(out)))

    def sm_mrq(self, k, inp, count, out):
        _check(lib().sealhip_sm_mrq(self.handle, k, _ptr(inp), count, _ptr(out)))

    def fast_floor(self, k, inp, count, out):
        _check(lib().sealhip_fast_floor(self.handle, k, _ptr(inp), count, _ptr(out)))

    def fastbconv_sk(self, k, inp, count, out):
        _check(lib().sealhip_fastbconv_sk(self.handle, k, _ptr(inp), count, _ptr(out)))

    def divide_and_round_q_last_inplace(self, k, data, count):
        _check(lib().sealhip_divide_and_round_q_last_inplace(self.handle, k, _ptr(data), count))

    def divide_and_round_q_last_ntt_inplace(self, k, data, count):
        _check(lib().sealhip_divide_and_round_q_last_ntt_inplace(self.handle, k, _ptr(data), count))

    def apply_galois(self, inp, count, k, galois_elt, out):
        _check(lib().sealhip_apply_galois(self.handle, _ptr(inp), count, k, galois_elt, _ptr(out)))

    def apply_galois_ntt(self, inp, count, k, galois_elt, out):
        _check(lib().sealhip_apply_galois_ntt(self.handle, _ptr(inp), count, k, galois_elt, _ptr(out)))

    def modup_rns(self, k, src_bundle_index, ext, count):
        _check(lib().sealhip_modup_rns(self.handle, k, src_bundle_index, _ptr(ext), count))

    def rescale_special_rns_inplace(self, k, poly, count):
        _check(lib().sealhip_rescale_special_rns_inplace(self.handle, k, _ptr(poly), count))

    def switch_key_inplace(self, k, ct, target, count, key):
        _check(lib().sealhip_switch_key_inplace(self.handle, k, _ptr(ct), _ptr(target), count, key.handle))

    # ---- SURVEY 8(e) latency mode: the digits of one key switch split across devices
    def kswitch_digits(self, k):
        d = C.c_uint32()
        _check(lib().sealhip_kswitch_digits(self.handle, k, C.byref(d)))
        return d.value

    def chunk_log(self):
        """[(batch size, items per arena chunk), ...] of this thread's last operations (sealhip_debug_chunk_log); clears it"""
        buf = (C.c_size_t * 128)()
        n = C.c_size_t(0)
        _check(lib().sealhip_debug_chunk_log(self.handle, buf, 64, C.byref(n)))
        return [(int(buf[2 * i]), int(buf[2 * i + 1])) for i in range(n.value)]

    def butterfly_rate(self, kind, prime_index=0):
        """butterflies per second of one butterfly sequence on this device (sealhip_debug_butterfly_rate)"""
        d = C.c_double(0.0)
        _check(lib().sealhip_debug_butterfly_rate(self.handle, kind, prime_index, C.byref(d)))
        return d.value

    def switch_key_partial(self, k, target, count, key, digit_begin, digit_end, partial):
        """inner product of evaluator.cpp:2302-2349 over the digits [digit_begin, digit_end) -> count x 2 x (k+nsp) x N"""
        _check(lib().sealhip_switch_key_partial(self.handle, k, _ptr(target), count, key.handle, digit_begin, digit_end,
                                                _ptr(partial)))

    def switch_key_finish(self, k, ct, partial_sum, count, n_partials):
        """the rest of the key switch (evaluator.cpp:2351-2366) on the element-wise sum of n_partials devices' partials"""
        _check(lib().sealhip_switch_key_finish(self.handle, k, _ptr(ct), _ptr(partial_sum), count, n_partials))

    def close(self):
        if getattr(self, "handle", None):
            lib().sealhip_context_destroy(self.handle)
            self.handle = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Evaluator:
    """Batched mirror of seal::Evaluator for the hot path (native/src/seal/evaluator.h:246-1239).

    Ciphertexts are device buffers in the reference layout (size x k x N uint64), `count` of them back to back;
    `k` names the level (number of coefficient-modulus primes)."""

    def __init__(self, context):
        self.ctx = context

    def multiply(self, a, size_a, b, size_b, k, count, out):
        _check(lib().sealhip_evaluator_multiply(self.ctx.handle, k, _ptr(a), size_a, _ptr(b), size_b, count, _ptr(out)))

    def square(self, a, size_a, k, count, out):
        _check(lib().sealhip_evaluator_square(self.ctx.handle, k, _ptr(a), size_a, count, _ptr(out)))

    def relinearize_inplace(self, ct, size, k, count, relin_keys):
        keys = (C.c_void_p * max(1, len(relin_keys)))(*[rk.handle for rk in relin_keys])
        _check(lib().sealhip_evaluator_relinearize(self.ctx.handle, k, _ptr(ct), size, count, keys, len(relin_keys)))

    def mod_switch_to_next(self, ct, size, k, count, out, item_stride=0):
        """item_stride (words): the ciphertexts of `ct` sit that far apart (e.g. the size-2 result of relinearize inside its
        size-3 product); 0 = back to back"""
        if item_stride:
            _check(lib().sealhip_evaluator_mod_switch_to_next_strided(self.ctx.handle, k, _ptr(ct), size, item_stride, count,
                                                                     _ptr(out)))
        else:
            _check(lib().sealhip_evaluator_mod_switch_to_next(self.ctx.handle, k, _ptr(ct), size, count, _ptr(out)))

    def rescale_to_next(self, ct, size, k, count, out, item_stride=0):
        if item_stride:
            _check(lib().sealhip_evaluator_rescale_to_next_strided(self.ctx.handle, k, _ptr(ct), size, item_stride, count,
                                                                  _ptr(out)))
        else:
            _check(lib().sealhip_evaluator_rescale_to_next(self.ctx.handle, k, _ptr(ct), size, count, _ptr(out)))

    def apply_galois_inplace(self, ct, k, count, galois_elt, galois_key):
        _check(lib().sealhip_evaluator_apply_galois(self.ctx.handle, k, _ptr(ct), count, galois_elt, galois_key.handle))

    def rotate_vector_inplace(self, ct, k, count, steps, galois_keys):
        """rotate_internal (evaluator.cpp:1945-2000): direct key if present, else the NAF decomposition.
        galois_keys: dict galois_elt -> KSwitchKeys."""
        if steps == 0:
            return
        elt = self.ctx.galois_elt_from_step(steps)
        if elt in galois_keys:
            return self.apply_galois_inplace(ct, k, count, elt, galois_keys[elt])
        naf = _naf(steps)
        if len(naf) == 1:
            raise ValueError("Galois key not present")
        for s in naf:
            if abs(s) != (self.ctx.n >> 1):
                self.rotate_vector_inplace(ct, k, count, s, galois_keys)

    rotate_rows_inplace = rotate_vector_inplace

    def rotate_vector_native(self, ct, k, count, steps, galois_keys):
        """The same rotate_internal logic behind the C ABI (sealhip_evaluator_rotate_vector)."""
        elts = list(galois_keys.keys())
        ea = (_u32 * max(1, len(elts)))(*elts)
        ka = (_vp * max(1, len(elts)))(*[galois_keys[g].handle for g in elts])
        _check(lib().sealhip_evaluator_rotate_vector(self.ctx.handle, k, _ptr(ct), count, steps, ea, ka, len(elts)))

    # ---- batches of separately allocated HOST ciphertexts (lists of numpy arrays: what a vector<Ciphertext> is)
    @staticmethod
    def _host_ptrs(arrays):
        for a in arrays:
            assert a.dtype == np.uint64 and a.flags["C_CONTIGUOUS"]
        return (C.c_void_p * max(1, len(arrays)))(*[a.ctypes.data for a in arrays])

    def host_register(self, array):
        """sealhip_host_register: pins the array's memory in place; *_host entries then copy it without staging"""
        _check(lib().sealhip_host_register(self.ctx.handle, array.ctypes.data, array.nbytes))

    def host_unregister(self, array):
        _check(lib().sealhip_host_unregister(self.ctx.handle, array.ctypes.data))

    def multiply_host(self, a, size_a, b, size_b, k, out, relin_keys=None):
        """Evaluator::multiply (+ relinearize when relin_keys is given) over lists of host ciphertexts; out[i] is written"""
        keys = (C.c_void_p * max(1, len(relin_keys or [])))(*[rk.handle for rk in (relin_keys or [])])
        _check(lib().sealhip_evaluator_multiply_host(self.ctx.handle, k, self._host_ptrs(a), size_a, self._host_ptrs(b), size_b,
                                                     len(a), self._host_ptrs(out), keys if relin_keys else None,
                                                     len(relin_keys or [])))

    def relinearize_host(self, cts, size, k, relin_keys):
        keys = (C.c_void_p * max(1, len(relin_keys)))(*[rk.handle for rk in relin_keys])
        _check(lib().sealhip_evaluator_relinearize_host(self.ctx.handle, k, self._host_ptrs(cts), size, len(cts), keys,
                                                        len(relin_keys)))

    def rotate_vector_host(self, cts, k, steps, galois_keys):
        elts = list(galois_keys.keys())
        ea = (_u32 * max(1, len(elts)))(*elts)
        ka = (_vp * max(1, len(elts)))(*[galois_keys[g].handle for g in elts])
        _check(lib().sealhip_evaluator_rotate_vector_host(self.ctx.handle, k, self._host_ptrs(cts), len(cts), steps, ea, ka,
                                                          len(elts)))

    def mod_switch_to_next_host(self, cts, size, k, out, rescale=False):
        fn = lib().sealhip_evaluator_rescale_to_next_host if rescale else lib().sealhip_evaluator_mod_switch_to_next_host
        _check(fn(self.ctx.handle, k, self._host_ptrs(cts), size, len(cts), self._host_ptrs(out)))

    # ---- the rest of the Evaluator surface on device-resident batches
    def negate(self, ct, size, k, count, out):
        """Evaluator::negate (evaluator.cpp:65-88)"""
        _check(lib().sealhip_evaluator_negate(self.ctx.handle, k, _ptr(ct), size, count, _ptr(out)))

    def add(self, a, size_a, b, size_b, k, count, out):
        """Evaluator::add (evaluator.cpp:90-151)"""
        _check(lib().sealhip_evaluator_add(self.ctx.handle, k, _ptr(a), size_a, _ptr(b), size_b, count, _ptr(out)))

    def sub(self, a, size_a, b, size_b, k, count, out):
        """Evaluator::sub (evaluator.cpp:174-233)"""
        _check(lib().sealhip_evaluator_sub(self.ctx.handle, k, _ptr(a), size_a, _ptr(b), size_b, count, _ptr(out)))

    def multiply_plain_inplace(self, ct, size, k, count, plain, plain_stride=0, ntt_form=None):
        """Evaluator::multiply_plain_inplace (evaluator.cpp:1438-1473): NTT form -> multiply_plain_ntt (plain = k x N in
        NTT form), coefficient form (BFV) -> multiply_plain_normal (plain = N coefficients < t)."""
        if ntt_form is None:
            ntt_form = self.ctx.scheme == SCHEME_CKKS
        fn = lib().sealhip_evaluator_multiply_plain_ntt if ntt_form else lib().sealhip_evaluator_multiply_plain
        _check(fn(self.ctx.handle, k, _ptr(ct), size, count, _ptr(plain), plain_stride))

    def add_plain_inplace(self, ct, size, k, count, plain, plain_stride=None, subtract=False):
        """Evaluator::add_plain_inplace / sub_plain_inplace (evaluator.cpp:1290-1435): BFV plain = N coefficients < t,
        CKKS plain = k x N in NTT form"""
        if plain_stride is None:
            plain_stride = self.ctx.n if self.ctx.scheme == SCHEME_BFV else k * self.ctx.n
        _check(lib().sealhip_evaluator_add_plain(self.ctx.handle, k, _ptr(ct), size, count, _ptr(plain), plain_stride,
                                                 1 if subtract else 0))

    def sub_plain_inplace(self, ct, size, k, count, plain, plain_stride=None):
        self.add_plain_inplace(ct, size, k, count, plain, plain_stride, subtract=True)

    def resize(self, src, src_size, dst_size, k, count, dst=None):
        """Ciphertext::resize (ciphertext.cpp:84-124) over a batch; returns the destination buffer"""
        if dst is None:
            dst = self.ctx.alloc(count * dst_size * k * self.ctx.n)
        _check(lib().sealhip_ciphertext_resize(self.ctx.handle, k, _ptr(src), src_size, _ptr(dst), dst_size, count))
        return dst

    def add_many(self, cts, size, k, count, out):
        """Evaluator::add_many (evaluator.cpp:153-172): out = cts[0] + cts[1] + ... (left to right)"""
        if not cts:
            raise ValueError("encrypteds cannot be empty")
        self.add(cts[0], size, cts[1], size, k, count, out) if len(cts) > 1 else self.resize(cts[0], size, size, k, count, out)
        for c in cts[2:]:
            self.add(out, size, c, size, k, count, out)

    def multiply_many(self, cts, k, count, relin_keys):
        """Evaluator::multiply_many (evaluator.cpp:1180-1255), BFV, size-2 operands, behind the C ABI
        (sealhip_evaluator_multiply_many): the reference's queue order, each product relinearized. Returns the device buffer
        of the result ([count][2][k][N])."""
        out = self.ctx.alloc(count * 2 * k * self.ctx.n)
        ptrs = (C.c_void_p * max(1, len(cts)))(*[_ptr(c) for c in cts])
        keys = (C.c_void_p * max(1, len(relin_keys)))(*[rk.handle for rk in relin_keys])
        _check(lib().sealhip_evaluator_multiply_many(self.ctx.handle, k, ptrs, len(cts), count, keys, len(relin_keys), _ptr(out)))
        return out

    def exponentiate(self, ct, exponent, k, count, relin_keys):
        """Evaluator::exponentiate_inplace (evaluator.cpp:1257-1288) behind the C ABI (sealhip_evaluator_exponentiate)"""
        out = self.ctx.alloc(count * 2 * k * self.ctx.n)
        keys = (C.c_void_p * max(1, len(relin_keys)))(*[rk.handle for rk in relin_keys])
        _check(lib().sealhip_evaluator_exponentiate(self.ctx.handle, k, _ptr(ct), int(exponent), count, keys, len(relin_keys),
                                                    _ptr(out)))
        return out

    def mod_switch_to(self, ct, size, k, k_target, count):
        """Evaluator::mod_switch_to_inplace (evaluator.cpp:1038-1088): mod_switch_to_next until level k_target"""
        if k_target > k:
            raise ValueError("cannot switch to higher level modulus")
        cur = ct
        while k > k_target:
            nxt = self.ctx.alloc(count * size * (k - 1) * self.ctx.n)
            self.mod_switch_to_next(cur, size, k, count, nxt)
            cur, k = nxt, k - 1
        return cur

    def rescale_to(self, ct, size, k, k_target, count):
        """Evaluator::rescale_to_inplace (evaluator.cpp:1128-1178), CKKS"""
        if k_target > k:
            raise ValueError("cannot switch to higher level modulus")
        cur = ct
        while k > k_target:
            nxt = self.ctx.alloc(count * size * (k - 1) * self.ctx.n)
            self.rescale_to_next(cur, size, k, count, nxt)
            cur, k = nxt, k - 1
        return cur

    def check_not_transparent(self, ct, size, k, count):
        """evaluator.cpp:265-271 (SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT)"""
        if self.ctx.is_transparent(ct, size, k, count).any():
            raise LogicError("result ciphertext is transparent")

    def transform_to_ntt_inplace(self, ct, size, k, count):
        _check(lib().sealhip_evaluator_transform_to_ntt(self.ctx.handle, k, _ptr(ct), size, count))

    def transform_from_ntt_inplace(self, ct, size, k, count):
        _check(lib().sealhip_evaluator_transform_from_ntt(self.ctx.handle, k, _ptr(ct), size, count))


def _naf(value):
    """util/numth.h:22-42"""
    res, sign, value, i = [], value < 0, abs(value), 0
    while value:
        zi = 2 - (value % 4) if value % 2 else 0
        value = (value - zi) // 2
        if zi:
            res.append((-zi if sign else zi) * (1 << i))
        i += 1
    return res
