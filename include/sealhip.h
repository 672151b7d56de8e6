/*
 * sealhip.h -- C ABI of the MI355X-native RNS-NTT polynomial engine that replaces the hot path
 * behind seal::Evaluator (Gemini-SEAL, a Microsoft SEAL 3.5.3 fork).
 *
 * Conventions (modelled on the reference's own C export layer, native/src/seal/c/defines.h:34-58):
 *   - every function returns an HRESULT-compatible `long`: S_OK (0) or one of the SEALHIP_E_* codes;
 *     nothing throws across the ABI; sealhip_last_error_string() returns the thread's last message;
 *   - plain pointers and sizes only, no C++/torch types;
 *   - all polynomial data are `uint64_t` matrices in the reference layout: a ciphertext is `size`
 *     polynomials, each a row-major (k x N) matrix, RNS row i of polynomial j at
 *     data + (j*k + i)*N   (native/src/seal/ciphertext.h:359-368, util/iterator.h:746-766);
 *   - a *batch* is `count` such objects stored back to back (the data-parallel axis);
 *   - unless a function name ends in `_host`, every data pointer is a DEVICE pointer (hipMalloc'd,
 *     or a torch CUDA tensor's data_ptr()); work is enqueued on a HIP stream and is asynchronous
 *     until sealhip_synchronize();
 *   - threading: like seal::Evaluator (native/src/seal/evaluator.h:1375-1377) a context is re-entrant.
 *     Every host thread that calls into a context works on its own LANE (HIP stream + temporaries
 *     arena), so operations issued by different threads overlap on the device; operations issued by
 *     one thread run in order. Work of different threads is NOT ordered against each other: hand a
 *     buffer from one thread to another only after sealhip_synchronize() (which waits for every lane);
 *   - a device-side failure (the forward NTT's bounded hand-off wait timing out) is sticky: the next
 *     entry point that makes results host-visible (sealhip_synchronize, sealhip_memcpy_d2h,
 *     sealhip_ciphertext_save, sealhip_is_transparent, sealhip_is_data_valid_for,
 *     sealhip_profile_fetch, the *_host batch entries) returns SEALHIP_E_UNEXPECTED;
 *   - a ciphertext level is addressed by `k` = number of leading coefficient-modulus primes
 *     (the chain drops the last prime per level: native/src/seal/context.cpp:423-431).
 *
 * There is no CPU fallback: without a HIP device every compute entry point fails with
 * SEALHIP_E_UNEXPECTED.
 */
#ifndef SEALHIP_H
#define SEALHIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* HRESULT values of native/src/seal/c/defines.h:34-49 */
#define SEALHIP_S_OK 0L
#define SEALHIP_E_POINTER ((long)0x80004003L)
#define SEALHIP_E_INVALIDARG ((long)0x80070057L)
#define SEALHIP_E_OUTOFMEMORY ((long)0x8007000EL)
#define SEALHIP_E_UNEXPECTED ((long)0x8000FFFFL)
#define SEALHIP_COR_E_INVALIDOPERATION ((long)0x80131509L)

/* scheme_type values (native/src/seal/encryptionparams.h:24-34) */
#define SEALHIP_SCHEME_BFV 1u
#define SEALHIP_SCHEME_CKKS 2u

/* PARITY reproduces the reference bit for bit (including SURVEY F2/F3); STRICT is the
   mathematically correct variant (Harvey-corrected butterflies, NTT'd in-bundle rows for BFV). */
#define SEALHIP_MODE_PARITY 0u
#define SEALHIP_MODE_STRICT 1u

/* which RNS base the rows of a polynomial belong to */
#define SEALHIP_BASE_Q 0u   /* the first k coefficient-modulus primes                     */
#define SEALHIP_BASE_BSK 1u /* BEHZ auxiliary base Bsk of level k (B..., m_sk last)       */
#define SEALHIP_BASE_KEY 2u /* k ciphertext primes followed by the nsp special primes     */

typedef struct sealhip_context sealhip_context;
typedef struct sealhip_kswitch_key sealhip_kswitch_key;

/* Plain mirror of what the path needs from EncryptionParameters / SEALContext
   (native/src/seal/encryptionparams.h:205-214,319-322; native/src/seal/context.cpp:455-540). */
typedef struct sealhip_params
{
    uint32_t scheme;           /* SEALHIP_SCHEME_*                                                   */
    uint32_t log_n;            /* log2(poly_modulus_degree), 3..16                                   */
    uint32_t n_key_moduli;     /* |coeff_modulus| at key level (special primes last)                 */
    uint32_t n_special_primes; /* EncryptionParameters::n_special_primes(), >= 1                     */
    const uint64_t *key_moduli;
    uint64_t plain_modulus;    /* BFV only; 0 for CKKS                                               */
    uint32_t mode;             /* SEALHIP_MODE_*                                                     */
    int32_t device;            /* HIP device ordinal; -1 = host-only context (tables, no compute)    */
} sealhip_params;

/* ---------------------------------------------------------------- library / context */
const char *sealhip_last_error_string(void);
long sealhip_num_devices(int32_t *count);
long sealhip_context_create(const sealhip_params *params, sealhip_context **out);
long sealhip_context_destroy(sealhip_context *ctx);
long sealhip_context_first_level(const sealhip_context *ctx, uint32_t *k_first); /* = n_key - nsp */
long sealhip_context_bsk_size(sealhip_context *ctx, uint32_t k, uint32_t *bsk_size); /* |Bsk| of level k */
/* Stream of the CALLING THREAD's lane: a caller-owned hipStream_t (e.g. torch.cuda.Stream().cuda_stream), or NULL to go
   back to a private non-blocking stream. The legacy default stream cannot be named by its handle (it is NULL too): use
   sealhip_use_default_stream for it. */
long sealhip_set_stream(sealhip_context *ctx, void *hip_stream);
long sealhip_use_default_stream(sealhip_context *ctx);
/* waits for the work of every lane of the context (all host threads) and reports a pending device-side failure */
long sealhip_synchronize(sealhip_context *ctx);
/* introspection: lanes (per-thread stream + arena) the context has created so far */
long sealhip_context_lane_count(sealhip_context *ctx, uint32_t *lanes);

/* device memory helpers for hosts that do not bring their own allocator */
long sealhip_malloc(sealhip_context *ctx, size_t bytes, void **dptr);
long sealhip_free(sealhip_context *ctx, void *dptr);
long sealhip_memcpy_h2d(sealhip_context *ctx, void *dst_dev, const void *src_host, size_t bytes);
long sealhip_memcpy_d2h(sealhip_context *ctx, void *dst_host, const void *src_dev, size_t bytes);

/* Per-kernel timing with HIP events recorded on the launch stream. While enabled every kernel launch of
   the engine is bracketed by two events; sealhip_profile_fetch synchronises the stream, writes a JSON
   object {"<kernel tag>": {"launches": n, "ms": total, "units": u}, ...} (NUL-terminated) into `json`,
   and clears the records. `units` counts RNS rows for the NTT passes (0 for other kernels). */
long sealhip_profile_enable(sealhip_context *ctx, int32_t enable);
long sealhip_profile_fetch(sealhip_context *ctx, char *json, size_t capacity);

/* Test hook of the forward NTT's sibling hand-off (csrc/ntt.hip): spin_limit = polls before a waiting wave gives up
   (0 restores the default 2^24), suppress_signal != 0 withholds the "finished reading" signal so that every wait times
   out. Used by the tests to prove that the failure surfaces at every host-visible synchronisation point. */
long sealhip_debug_ntt_handoff(sealhip_context *ctx, uint32_t spin_limit, int32_t suppress_signal);

/* How the calling thread's last operations walked their batches: operations whose temporaries do not fit the lane's arena for
   the whole batch process it in chunks of items. Writes up to capacity_pairs (batch size, items per
   chunk) pairs, oldest first, and clears the log (at most the last 64 operations are kept). bench.py uses it to verify the
   first and last item of every chunk of the timed batch against the oracle. */
long sealhip_debug_chunk_log(sealhip_context *ctx, size_t *count_chunk_pairs, size_t capacity_pairs, size_t *written);

/* Measurement hook (bench.py roofline.valu_ceiling): the rate at which this device executes nothing but the butterflies of
   the single-pass NTT kernels (same instruction sequences, values and twiddles in registers, same launch bounds) on the
   prime `prime_index` (numbering of sealhip_debug_ntt_table). kind: 0 the reference's lazy forward butterfly
   (ntt.cpp:245-252, exact Shoup quotient), 1 / 2 the approximate-quotient forms (csrc/ntt_bounds.hpp section 2), 3 the
   FP64 butterfly (primes below 2^50), 4 / 5 the inverse butterfly (lazy sums with the level-2 quotient / ntt.cpp:265-272).
   A row of N coefficients is N/2 log2 N butterflies: rate / that = the transform's arithmetic ceiling in rows per second. */
long sealhip_debug_butterfly_rate(sealhip_context *ctx, uint32_t kind, uint32_t prime_index, double *butterflies_per_s);

/* Introspection of the precomputed tables (works on host-only contexts; used by the CPU tests).
   kind: 0 root_powers, 1 scaled_root_powers, 2 inv_root_powers (reference order, n^-1 merged),
   3 scaled_inv_root_powers; prime_index: 0..n_key-1 key primes, n_key.. = 60-bit auxiliary primes
   in get_primes order (m_sk, gamma, B_0, B_1, ...). */
long sealhip_debug_ntt_table(sealhip_context *ctx, uint32_t prime_index, uint32_t kind, uint64_t *out_host,
                             size_t capacity);
/* which: 0 Bsk primes, 1 inv_prod_q_mod_Bsk, 2 prod_q_mod_Bsk, 3 inv_m_tilde_mod_Bsk, 4 prod_B_mod_q,
   5 inv_q_last_mod_q, 6 {inv_prod_q_mod_m_tilde, inv_prod_B_mod_m_sk, m_sk, gamma},
   7 q->Bsk matrix (row-major [Bsk][q]), 8 B->q matrix ([q][B]), 9 q inv_punctured, 10 B inv_punctured,
   11 q->m_tilde row, 12 B->m_sk row */
long sealhip_debug_rns_constants(sealhip_context *ctx, uint32_t k, uint32_t which, uint64_t *out_host,
                                 size_t capacity, size_t *written);

/* ---------------------------------------------------------------- L2: NTT (util/ntt.h:189-368)
   data: count polynomials x rows x N, in place. `base` selects the primes of the `rows` rows of one
   polynomial: BASE_Q -> rows = k; BASE_BSK -> rows = |Bsk|(k); BASE_KEY -> rows = k + nsp.
   Operand ranges are what the reference's butterflies are written for (ntt.cpp:245-281, :341): forward inputs below 4p, inverse inputs below 2p.
   The `_lazy` entries reproduce the reference's representatives word for word (including the wrapped words of the
   60-bit Bsk rows). The canonicalising entries return the residues those words reduce to; on rows whose prime
   is below 2^50 they are computed with exact double-precision butterflies, which for operands
   inside the ranges above is the same function -- for words outside them (>= 2^52) both the reference's output and this
   one are meaningless, and they differ. Likewise the canonicalising forward entry on primes below 2^58 runs a cheaper
   exact schedule (approximate Shoup quotients, one reduction in the store) that is proved for inputs below 4p
   (csrc/ntt_bounds.hpp: fwd_canon_admits); SEALHIP_NTT_CANON_EXACT=1 runs the reference's own sequence instead, whose
   deterministic wrap-around on larger words is then reproduced as well. */
long sealhip_ntt_negacyclic_harvey_lazy(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k,
                                        uint32_t base);
long sealhip_ntt_negacyclic_harvey(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k, uint32_t base);
long sealhip_inverse_ntt_negacyclic_harvey_lazy(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k,
                                                uint32_t base);
long sealhip_inverse_ntt_negacyclic_harvey(sealhip_context *ctx, uint64_t *data, size_t count, uint32_t k,
                                           uint32_t base);

/* ---------------------------------------------------------------- L2: coefficient-wise (util/polyarithsmallmod.{h,cpp})
   operands: count polynomials x rows x N of the given base/level; result may alias an operand. */
long sealhip_dyadic_product_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count,
                                     uint32_t k, uint32_t base, uint64_t *result);
long sealhip_multiply_poly_scalar_coeffmod(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k,
                                           uint32_t base, uint64_t scalar, uint64_t *result);
long sealhip_add_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count, uint32_t k,
                               uint32_t base, uint64_t *result);
long sealhip_sub_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, const uint64_t *b, size_t count, uint32_t k,
                               uint32_t base, uint64_t *result);
long sealhip_negate_poly_coeffmod(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k, uint32_t base,
                                  uint64_t *result);

/* ---------------------------------------------------------------- L2: RNSTool (util/rns.cpp:731-1068), level k, batched
   shapes per item:  fastbconv_m_tilde k x N -> (|Bsk|+1) x N;  sm_mrq (|Bsk|+1) x N -> |Bsk| x N;
   fast_floor (k+|Bsk|) x N -> |Bsk| x N;  fastbconv_sk |Bsk| x N -> k x N;
   divide_and_round_q_last[_ntt]_inplace: k x N in place (last row clobbered). */
long sealhip_fastbconv_m_tilde(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out);
long sealhip_sm_mrq(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out);
long sealhip_fast_floor(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out);
long sealhip_fastbconv_sk(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out);
long sealhip_divide_and_round_q_last_inplace(sealhip_context *ctx, uint32_t k, uint64_t *data, size_t count);
long sealhip_divide_and_round_q_last_ntt_inplace(sealhip_context *ctx, uint32_t k, uint64_t *data, size_t count);

/* ---------------------------------------------------------------- L2: Galois (util/galois.cpp) */
long sealhip_galois_elt_from_step(const sealhip_context *ctx, int32_t step, uint32_t *galois_elt);
/* coefficient form (galois.cpp:144-186) / NTT form (:188-214); in and out must not alias */
long sealhip_apply_galois(sealhip_context *ctx, const uint64_t *in, size_t count, uint32_t k, uint32_t galois_elt,
                          uint64_t *out);
long sealhip_apply_galois_ntt(sealhip_context *ctx, const uint64_t *in, size_t count, uint32_t k,
                              uint32_t galois_elt, uint64_t *out);

/* ---------------------------------------------------------------- hybrid key switch (multi_special_primes.cpp, evaluator.cpp:2259-2368) */
/* Key in the K1 layout of KeyGenerator::generate_one_kswitch_key (keygenerator.cpp:325-369):
   n_digits x 2 x n_key x N uint64 (digit, component, key-level row, coefficient). The key is copied
   to the device (from host memory if from_host != 0) and stays resident until destroyed. */
long sealhip_kswitch_key_load(sealhip_context *ctx, const uint64_t *key, uint32_t n_digits, int32_t from_host,
                              sealhip_kswitch_key **out);
long sealhip_kswitch_key_destroy(sealhip_context *ctx, sealhip_kswitch_key *key);
/* modup_rns (multi_special_primes.cpp:151-185): ext is count x (k+nsp) x N; the rows of bundle
   `src_bundle_index` are the input, every other row is overwritten. */
long sealhip_modup_rns(sealhip_context *ctx, uint32_t k, uint32_t src_bundle_index, uint64_t *ext, size_t count);
/* rescale_special_rns_inplace (multi_special_primes.cpp:237-304): poly is count x (k+nsp) x N */
long sealhip_rescale_special_rns_inplace(sealhip_context *ctx, uint32_t k, uint64_t *poly, size_t count);
/* switch_key_inplace: ct is count x 2 x k x N (updated in place), target count x k x N */
long sealhip_switch_key_inplace(sealhip_context *ctx, uint32_t k, uint64_t *ct, const uint64_t *target, size_t count,
                                const sealhip_kswitch_key *key);
/* switch_key_inplace in "latency mode" (SURVEY 8e): the d = ceil(k / nsp) decomposition digits (keygenerator.cpp:334-336;
   sealhip_kswitch_digits) of ONE key switch shared out over several devices, the key replicated on each. Every device calls
   _partial for its digits [digit_begin, digit_end): mod-up, forward transforms and the 128-bit inner product of
   evaluator.cpp:2302-2349 over those digits only, reduced to canonical residues -> partial = count x 2 x (k + nsp) x N words.
   The caller adds the partials of all devices element-wise (an all-reduce with SUM on 64-bit words: ranks * p < 2^63 cannot
   wrap) and hands the sum to _finish on the device(s) that need the result: one more reduction, then :2351-2366 as in the
   unsplit operation (partial_sum is clobbered). Modular sums are associative, so the result is word for word the unsplit one.
   n_partials = how many partials were summed (the world size of the all-reduce): _finish reduces with barrett_reduce_63 and
   returns E_INVALIDARG when n_partials * max(key prime) could reach 2^63 (61-bit primes: more than 4). Measured with one
   device only; the multi-device run is tools/latency_mode.py under torch.distributed. */
long sealhip_switch_key_partial(sealhip_context *ctx, uint32_t k, const uint64_t *target, size_t count,
                                const sealhip_kswitch_key *key, uint32_t digit_begin, uint32_t digit_end, uint64_t *partial);
long sealhip_switch_key_finish(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint64_t *partial_sum, size_t count,
                               uint32_t n_partials);
long sealhip_kswitch_digits(sealhip_context *ctx, uint32_t k, uint32_t *digits);

/* ---------------------------------------------------------------- L4: Evaluator operations (native/src/seal/evaluator.h)
   All are batched over `count` independent ciphertexts at level k. */
/* Evaluator::multiply (evaluator.cpp:235-527): a (size_a polys), b (size_b polys) -> out (size_a+size_b-1).
   BFV: coefficient form in/out (bfv_multiply); CKKS: NTT form (ckks_multiply). out must not alias a or b. */
long sealhip_evaluator_multiply(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a,
                                const uint64_t *b, uint32_t size_b, size_t count, uint64_t *out);
/* Evaluator::square (evaluator.cpp:529-770) */
long sealhip_evaluator_square(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, size_t count,
                              uint64_t *out);
/* Evaluator::relinearize (evaluator.cpp:772-827): ct has `size` polys and is reduced in place to 2;
   relin_keys[i] is the key at RelinKeys::get_index(i + 2) (relinkeys.h:61-68). The batch stride stays
   size x k x N. */
long sealhip_evaluator_relinearize(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                   const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys);
/* Evaluator::mod_switch_to_next (evaluator.cpp:996-1036): BFV divide-and-round, CKKS drop.
   out: count x size x (k-1) x N */
long sealhip_evaluator_mod_switch_to_next(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                          size_t count, uint64_t *out);
/* Evaluator::rescale_to_next (evaluator.cpp:1090-1126), CKKS only */
long sealhip_evaluator_rescale_to_next(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                       size_t count, uint64_t *out);
/* The same two operations on ciphertexts that sit `ct_item_stride` words apart (>= size * k * N): the size-2 result of
   relinearize inside its size-3 product, so that multiply -> relinearize -> mod_switch_to_next over a contiguous device batch
   needs no compaction pass in between. The reference's ciphertexts are separate objects (ciphertext.h:709-721); a contiguous
   batch takes the stride instead ("a contiguous batch with stride"). `out` is compact (size * (k-1) * N per item). */
long sealhip_evaluator_mod_switch_to_next_strided(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                                  size_t ct_item_stride, size_t count, uint64_t *out);
long sealhip_evaluator_rescale_to_next_strided(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size,
                                               size_t ct_item_stride, size_t count, uint64_t *out);
/* Evaluator::apply_galois_inplace (evaluator.cpp:1841-1943): ct is count x 2 x k x N */
long sealhip_evaluator_apply_galois(sealhip_context *ctx, uint32_t k, uint64_t *ct, size_t count,
                                    uint32_t galois_elt, const sealhip_kswitch_key *galois_key);
/* Evaluator::transform_to_ntt_inplace / transform_from_ntt_inplace (evaluator.cpp:1746-1839) */
long sealhip_evaluator_transform_to_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count);
long sealhip_evaluator_transform_from_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size,
                                          size_t count);

/* Evaluator::multiply_many (evaluator.cpp:1180-1255), BFV: encrypteds[0..n) are device batches of count size-2 ciphertexts
   ([count][2][k][N]); out receives their product, relinearized after every multiplication, in the reference's queue order
   (neighbours left to right, an odd last operand appended, products of products until one is left). relin_keys as in
   sealhip_evaluator_relinearize. out must not be one of the operands. Asynchronous (stream-ordered temporaries). */
long sealhip_evaluator_multiply_many(sealhip_context *ctx, uint32_t k, const uint64_t *const *encrypteds, uint32_t n_encrypteds,
                                     size_t count, const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys,
                                     uint64_t *out);
/* Evaluator::exponentiate_inplace (evaluator.cpp:1257-1288): multiply_many over `exponent` copies; E_INVALIDARG for 0. */
long sealhip_evaluator_exponentiate(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint64_t exponent, size_t count,
                                    const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys, uint64_t *out);

/* ---------------------------------------------------------------- batches of separately allocated HOST ciphertexts
   What the reference's objects look like from C: a std::vector<seal::Ciphertext> is one separately allocated buffer per
   ciphertext (Ciphertext::data() of each element; native/src/seal/ciphertext.h:327-392,709-721). These entries take arrays of
   HOST pointers, one per ciphertext, and pipeline the batch through the device in chunks (SEALHIP_HOST_CHUNK, default 64):
   host threads gather the caller's buffers into pinned staging memory, host->device copy, the operation and device->host
   copy run on three HIP streams with two staging slots, host threads scatter the results -- so PCIe traffic in both
   directions overlaps the kernels. The calls return when every result is in the caller's buffers (they synchronise and
   report device-side failures). The library never frees or reallocates caller memory: every out[i] must already have room.

   Evaluator::multiply (evaluator.cpp:235-527) per pair (a[i], b[i]) -> out[i]. With relin_keys != NULL the product is
   relinearized before it leaves the device (Evaluator::relinearize_inplace, :772-827; relin_keys as in
   sealhip_evaluator_relinearize) and out[i] receives 2 polynomials; otherwise size_a + size_b - 1. */
long sealhip_evaluator_multiply_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *a, uint32_t size_a,
                                     const uint64_t *const *b, uint32_t size_b, size_t count, uint64_t *const *out,
                                     const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys);
/* Evaluator::relinearize_inplace: ct[i] holds `size` polynomials on entry; its first 2 polynomials are the result (the
   caller shrinks its object, evaluator.cpp:819). */
long sealhip_evaluator_relinearize_host(sealhip_context *ctx, uint32_t k, uint64_t *const *ct, uint32_t size, size_t count,
                                        const sealhip_kswitch_key *const *relin_keys, uint32_t n_relin_keys);
/* Evaluator::rotate_vector_inplace / rotate_rows_inplace (evaluator.h:1201-1239) on size-2 ciphertexts, in place; keys as in
   sealhip_evaluator_rotate_vector. */
long sealhip_evaluator_rotate_vector_host(sealhip_context *ctx, uint32_t k, uint64_t *const *ct, size_t count, int32_t steps,
                                          const uint32_t *galois_elts, const sealhip_kswitch_key *const *galois_keys,
                                          uint32_t n_keys);
/* Evaluator::mod_switch_to_next / rescale_to_next: ct[i] (size x k x N) -> out[i] (size x (k-1) x N) */
long sealhip_evaluator_mod_switch_to_next_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *ct, uint32_t size,
                                               size_t count, uint64_t *const *out);
long sealhip_evaluator_rescale_to_next_host(sealhip_context *ctx, uint32_t k, const uint64_t *const *ct, uint32_t size,
                                            size_t count, uint64_t *const *out);

/* Pins the caller's range [ptr, ptr + bytes) in place (hipHostRegister, visible to every device) and remembers it. A *_host
   entry whose item buffers of an array all lie inside registered ranges skips that array's staging copy: the DMA engine reads /
   writes the caller's buffers themselves, which takes the host threads (the bound of the pageable path) out of the transfer.
   Meant for memory the caller keeps -- the reference's ciphertexts come from a MemoryPool whose blocks are allocated at
   native/src/seal/util/mempool.cpp:45 and :145 and reused for the life of the pool: register a block once where it is
   allocated, unregister it where the pool frees it. Pinning costs far more than one copy, so registering a buffer for a single
   call does not pay. E_INVALIDARG: empty range, a range that overlaps a registered one, (unregister) a pointer that is not the
   start of a registered range; HIP's refusal (e.g. a locked-memory limit) -> E_UNEXPECTED with its message. The caller must not
   unregister or free a range while a *_host call that uses it is running. */
long sealhip_host_register(sealhip_context *ctx, void *ptr, size_t bytes);
long sealhip_host_unregister(sealhip_context *ctx, void *ptr);

/* ---------------------------------------------------------------- Evaluator surface beyond the hot path */
/* Ciphertext batches [count][size][k][N]. Evaluator::negate_inplace (evaluator.cpp:65-88); out may alias ct. */
long sealhip_evaluator_negate(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                              uint64_t *out);
/* Evaluator::add_inplace (evaluator.cpp:90-151) / sub_inplace (:174-233): out has max(size_a, size_b) polynomials per
   ciphertext; the tail of the longer operand is copied (add) or, when it is b's, negated (sub, :216-220).
   out may alias a when size_a >= size_b. */
long sealhip_evaluator_add(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                           uint32_t size_b, size_t count, uint64_t *out);
long sealhip_evaluator_sub(sealhip_context *ctx, uint32_t k, const uint64_t *a, uint32_t size_a, const uint64_t *b,
                           uint32_t size_b, size_t count, uint64_t *out);
/* Evaluator::multiply_plain_ntt (evaluator.cpp:1605-1646): every polynomial of every ciphertext times a plaintext in
   NTT form (k x N), in place. plain_stride = words between the plaintexts of consecutive ciphertexts, 0 = one
   plaintext for the whole batch. */
long sealhip_evaluator_multiply_plain_ntt(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                          const uint64_t *plain_ntt, size_t plain_stride);
/* Evaluator::multiply_plain_normal (evaluator.cpp:1475-1603), BFV, coefficient form, in place. plain = N coefficients
   in [0, t) per plaintext (plain_stride as above). Requires every q_i > t ("fast plain lift", context.cpp:297-301);
   otherwise COR_E_INVALIDOPERATION. */
long sealhip_evaluator_multiply_plain(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                      const uint64_t *plain, size_t plain_stride);
/* Ciphertext::is_transparent (ciphertext.h:471-476) per ciphertext of the batch: transparent[i] = 1 when polynomials
   1.. are identically zero (or size < 2). `transparent` is HOST memory (count bytes); the call synchronises.
   The reference throws logic_error("result ciphertext is transparent") after every operation when built with
   SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT (evaluator.cpp:265-271); the adapter does the same with this flag. */
long sealhip_is_transparent(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                            uint8_t *transparent);
/* The same test as a FLAG OUTPUT of the operations ("provide it as a flag output of the kernels"; replaces the
   read pass of Ciphertext::is_transparent that evaluator.cpp:265-271 runs after every operation). `nonzero_flags` is DEVICE
   memory, `capacity` 32-bit words; it belongs to the calling thread's lane of the context until it is replaced or removed
   (nullptr). While a sink is set, every sealhip_evaluator_* entry that produces ciphertexts first clears flags[0 .. count) and
   then makes flags[i] non-zero iff polynomials 1.. of result i hold a non-zero word: multiply, square, relinearize and
   apply_galois note it in the kernel that stores those polynomials anyway (no pass over the result); mod_switch_to_next,
   rescale_to_next, rotate_vector, add, sub, negate and multiply_plain(_ntt) run the read pass on their result, on the stream.
   A batch larger than `capacity` is E_INVALIDARG. Nothing synchronises: read the flags after sealhip_synchronize (or on the
   lane's stream). Composite entries (multiply_many, exponentiate, the *_host batches) leave the flags alone. */
long sealhip_transparency_sink(sealhip_context *ctx, uint32_t *nonzero_flags, size_t capacity);
/* modulo_poly_coeffs_63 (polyarithsmallmod.h:98-120): Barrett-63 reduction of rows with values < 2^63 */
long sealhip_modulo_poly_coeffs_63(sealhip_context *ctx, const uint64_t *a, size_t count, uint32_t k, uint32_t base,
                                   uint64_t *result);
/* Evaluator::rotate_vector_inplace / rotate_rows_inplace -> rotate_internal (evaluator.cpp:1945-2000): the key for
   the step's Galois element if the caller holds it, else the non-adjacent-form decomposition (util/numth.h:22-42).
   galois_elts[i] is the Galois element of galois_keys[i]. */
long sealhip_evaluator_rotate_vector(sealhip_context *ctx, uint32_t k, uint64_t *ct, size_t count, int32_t steps,
                                     const uint32_t *galois_elts, const sealhip_kswitch_key *const *galois_keys,
                                     uint32_t n_keys);

/* ---------------------------------------------------------------- decrypt-side arithmetic */
/* Decryptor::dot_product_ct_sk_array (decryptor.cpp:218-265): out[count][k][N] = c_0 + sum_{i>=1} c_i * s^i, in the form
   of the ciphertext (is_ntt_form). sk_powers_ntt = the Decryptor's secret_key_array_: (size-1) polynomials s, s^2, ...
   in NTT form, each with the key level's row stride (n_key_moduli x N); device memory. */
long sealhip_decryptor_dot_product_ct_sk(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                                         const uint64_t *sk_powers_ntt, int32_t is_ntt_form, uint64_t *out);
/* RNSTool::decrypt_scale_and_round (rns.cpp:1070-1126), BFV: in[count][k][N] -> out[count][N] coefficients mod t */
long sealhip_decrypt_scale_and_round(sealhip_context *ctx, uint32_t k, const uint64_t *in, size_t count, uint64_t *out);

/* ---------------------------------------------------------------- HIP graphs for launch-bound small batches */
/* Small batches are bound by kernel launches (one multiply+relinearize of a single N=2^15 ciphertext is ~25 launches):
   a fixed sequence of operations on fixed device buffers can be captured once from the context's stream and replayed as
   one hipGraph launch. Run the sequence once before capturing (tables, the arena and the NTT tickets are allocated on
   first use; allocation and synchronisation are not capturable); entry points that synchronise (is_transparent,
   ckks_encode, the wire format) cannot be captured. An operation that fails during a capture aborts and discards the
   capture (the error message says so). A graph replays on the lane (thread) it was captured on, whichever thread
   launches it; it goes stale -- launch fails with COR_E_INVALIDOPERATION -- when that lane's workspace is re-allocated
   or any key-switch key of the context is destroyed. Operand buffers are the caller's: they must outlive the graph. */
typedef struct sealhip_graph sealhip_graph;
long sealhip_graph_capture_begin(sealhip_context *ctx);
long sealhip_graph_capture_end(sealhip_context *ctx, sealhip_graph **graph);
long sealhip_graph_launch(sealhip_context *ctx, sealhip_graph *graph);
long sealhip_graph_destroy(sealhip_context *ctx, sealhip_graph *graph);

/* ---------------------------------------------------------------- encrypt-side arithmetic */
/* util::encrypt_zero_symmetric (util/rlwe.cpp:204-300) with the random samples handed in (sampling and the CSPRNG
   stay on the host): ct[count][2][rows][N] = ([-(a*s + e)]_q, a) over key primes 0..rows-1 (rows = n_key_moduli for
   key generation, keygenerator.cpp:347; rows = the level's k for Encryptor::encrypt_zero_internal, encryptor.cpp:183).
   a_ntt[count][rows][N]: uniform residues, taken to be in NTT form as the reference samples them (:245-249);
   noise[count][N]: small signed error coefficients (sample_poly_normal, :61-95); sk_ntt: rows x N, NTT form.
   is_ntt_form != 0 leaves the ciphertext in NTT form (CKKS, keys), 0 in coefficient form (BFV). Device memory. */
long sealhip_encrypt_zero_symmetric(sealhip_context *ctx, uint32_t rows, int32_t is_ntt_form, const uint64_t *a_ntt,
                                    const int32_t *noise, const uint64_t *sk_ntt, size_t count, uint64_t *ct);
/* util::encrypt_zero_asymmetric (util/rlwe.cpp:140-202): ct[count][2][rows][N], ct_j = pk_j * u + e_j.
   pk_ntt[2][rows][N] (NTT form); u[count][N] ternary (sample_poly_ternary, :25-59); noise[count][2][N]. */
long sealhip_encrypt_zero_asymmetric(sealhip_context *ctx, uint32_t rows, int32_t is_ntt_form, const uint64_t *pk_ntt,
                                     const int32_t *u, const int32_t *noise, size_t count, uint64_t *ct);
/* util::multiply_add_plain_with_scaling_variant / multiply_sub_plain_with_scaling_variant
   (util/scalingvariant.cpp:15-52 / :54-92) on c_0 of every ciphertext: the last step of Encryptor::encrypt for BFV
   (encryptor.cpp:221-225) and Evaluator::add_plain_inplace / sub_plain_inplace for BFV (evaluator.cpp:1338-1342).
   plain[count][N] coefficients < t (plain_item_stride words apart; 0 = one plaintext for all);
   ct[count][size][k][N] in place. */
long sealhip_multiply_add_plain_with_scaling_variant(sealhip_context *ctx, uint32_t k, const uint64_t *plain,
                                                     size_t plain_item_stride, uint64_t *ct, uint32_t size, size_t count,
                                                     int32_t subtract);
/* Evaluator::add_plain_inplace / sub_plain_inplace (evaluator.cpp:1290-1435): BFV = the scaling variant above;
   CKKS = add/sub_poly_coeffmod of the NTT-form plaintext plain[count][k][N] on c_0. */
long sealhip_evaluator_add_plain(sealhip_context *ctx, uint32_t k, uint64_t *ct, uint32_t size, size_t count,
                                 const uint64_t *plain, size_t plain_item_stride, int32_t subtract);

/* ---------------------------------------------------------------- BatchEncoder */
/* 1 when the context can batch: BFV with a prime plain modulus = 1 (mod 2N) (context.cpp:262-275, qualifiers().using_batching) */
long sealhip_context_using_batching(const sealhip_context *ctx, int32_t *using_batching);
/* BatchEncoder::encode (batchencoder.cpp:113-154): values[count][n_values] (n_values <= N, each < t; missing slots are
   zero) -> plain[count][N] coefficients; BatchEncoder::decode (:339-376): plain[count][N] -> values[count][N].
   E_INVALIDARG when the parameters do not support batching. Device memory. */
long sealhip_batch_encode(sealhip_context *ctx, const uint64_t *values, size_t n_values, size_t count, uint64_t *plain);
long sealhip_batch_decode(sealhip_context *ctx, const uint64_t *plain, size_t count, uint64_t *values);
/* the vector<int64_t> overloads (batchencoder.cpp:156-198, :378-420): values in (-t/2, t/2]; a negative value is stored as
   t + v, a decoded slot above t/2 comes back as v - t */
long sealhip_batch_encode_int64(sealhip_context *ctx, const int64_t *values, size_t n_values, size_t count, uint64_t *plain);
long sealhip_batch_decode_int64(sealhip_context *ctx, const uint64_t *plain, size_t count, int64_t *values);

/* CKKSEncoder::encode (ckks.h:405-617) / decode (:623-747), double precision. values: complex numbers as (re, im) pairs
   of doubles in device memory. encode: values[count][n_values] (n_values <= N/2; the other slots are zero) -> plain
   [count][k][N] in NTT form at `scale`; decode: plain[count][k][N] -> values[count][N/2]. Every floating-point operation
   is issued in the reference's order without contraction, so the outputs equal the reference's bits given the same root
   tables (host libm). E_INVALIDARG: "scale out of bounds", "encoded values are too large", "values_size is too large". */
long sealhip_ckks_encode(sealhip_context *ctx, uint32_t k, const double *values, size_t n_values, size_t count, double scale,
                         uint64_t *plain);
long sealhip_ckks_decode(sealhip_context *ctx, uint32_t k, const uint64_t *plain, size_t count, double scale, double *values);
/* CKKSEncoder::encode(double value, ...) (ckks.cpp:80-216): `value` in every slot = the constant polynomial round(value*scale);
   plain[count][k][N] (the same plaintext `count` times), NTT form. Errors as the reference: "scale out of bounds", "encoded
   value is too large". sealhip_ckks_encode_int64: CKKSEncoder::encode(int64_t value, ...) (ckks.cpp:218-275), scale 1. */
long sealhip_ckks_encode_value(sealhip_context *ctx, uint32_t k, double value, double scale, size_t count, uint64_t *plain);
long sealhip_ckks_encode_int64(sealhip_context *ctx, uint32_t k, int64_t value, size_t count, uint64_t *plain);

/* Ciphertext::resize (ciphertext.cpp:84-124) over a device-resident batch: dst[count][dst_size][k][N] receives the first
   min(src_size, dst_size) polynomials of every src[count][src_size][k][N]; added polynomials are zero (IntArray::resize).
   What a chain needs between relinearize (which leaves the batch stride at its old size) and the next multiply. */
long sealhip_ciphertext_resize(sealhip_context *ctx, uint32_t k, const uint64_t *src, uint32_t src_size, uint64_t *dst,
                               uint32_t dst_size, size_t count);

/* ---------------------------------------------------------------- ciphertext wire format */
/* What Ciphertext::save_members writes ahead of the coefficient words (ciphertext.cpp:170-188). */
typedef struct sealhip_ciphertext_info
{
    uint64_t parms_id[4];         /* parms_id_type (4 x uint64 Blake2 hash, computed by the host library)  */
    uint32_t is_ntt_form;
    uint32_t size;                /* polynomials                                                           */
    uint32_t coeff_modulus_size;  /* k                                                                     */
    uint32_t seeded;              /* 1: c_1 was replaced by a PRNG seed (ciphertext.cpp:189-208)           */
    uint64_t poly_modulus_degree; /* N                                                                     */
    double scale;
    uint64_t data_words;          /* uint64 words stored in the stream                                     */
    uint64_t total_bytes;         /* SEALHeader::size of the whole object                                  */
} sealhip_ciphertext_info;
/* Registers the parms_id of level k (k = n_key_moduli: the key level). The binding copies them from
   SEALContext::get_context_data(...)->parms_id() once per context (context.cpp:455-540 builds the chain);
   the loader uses them the way is_metadata_valid_for does (valcheck.cpp:67-105). */
long sealhip_context_set_parms_id(sealhip_context *ctx, uint32_t k, const uint64_t parms_id[4]);
/* Serialization::LoadHeader + the metadata of Ciphertext::load_members, no context needed (serialization.cpp:137-176,
   ciphertext.cpp:248-267). Errors as the reference: bad magic/size/compression -> COR_E_INVALIDOPERATION
   ("loaded SEALHeader is invalid" / "incompatible version"), truncated input -> E_UNEXPECTED ("I/O error"). */
long sealhip_ciphertext_peek(const void *bytes, size_t len, sealhip_ciphertext_info *info);
/* Ciphertext::load (ciphertext.cpp:228-330, uncompressed stream): validates the metadata against the context and
   copies the coefficient words from `bytes` (host) straight into dst_device (capacity in words). A seeded ciphertext
   (info->seeded: one stored polynomial + a 64-byte seed, what Encryptor::encrypt_symmetric(...).save() writes) is expanded
   like Ciphertext::expand_seed does (:126-133): c_1 is re-sampled on the host from the seed with the reference's
   BlakePRNG (BLAKE2Xb, randomgen.cpp:63-73) and sample_poly_uniform (util/rlwe.cpp:101-129), re-implemented in
   csrc/blake2xb.cpp from the BLAKE2 specifications; dst_device receives both polynomials. */
long sealhip_ciphertext_load(sealhip_context *ctx, const void *bytes, size_t len, sealhip_ciphertext_info *info,
                             uint64_t *dst_device, size_t capacity_words);
/* Ciphertext::save_size(compr_mode_type::none) (ciphertext.cpp:135-168) and Ciphertext::save: the stream is written
   into `bytes` (host) with the coefficient words copied straight from src_device. */
long sealhip_ciphertext_save_size(const sealhip_context *ctx, uint32_t size, uint32_t k, size_t *bytes);
long sealhip_ciphertext_save(sealhip_context *ctx, const sealhip_ciphertext_info *info, const uint64_t *src_device,
                             void *bytes, size_t capacity, size_t *written);
/* KSwitchKeys::load (kswitchkeys.cpp:87-150; RelinKeys / GaloisKeys streams, uncompressed): loads keys_[index] -- RelinKeys:
   index = key_power - 2 (relinkeys.h:61-68), GaloisKeys: index = (galois_elt - 1) / 2 (galoiskeys.h:52-55) -- with its
   decomposition digits concatenated straight from the stream into HBM. *key = NULL when that slot is empty; n_slots (may
   be NULL) receives keys_.size(). Needs the key level's parms_id registered (k = n_key_moduli). Seeded digits (keys saved
   as Serializable<RelinKeys>) are expanded as in sealhip_ciphertext_load. */
long sealhip_kswitch_key_load_stream(sealhip_context *ctx, const void *bytes, size_t len, uint32_t index,
                                     sealhip_kswitch_key **key, uint64_t *n_slots);
/* Ciphertext::expand_seed on the host (works on host-only contexts): out_host[rows][N] = the words of c_1 for `seed`
   (random_seed_type: 8 x uint64) over the first `rows` key primes. sealhip_debug_blake2xb: the BLAKE2Xb function under it. */
long sealhip_expand_seed_host(sealhip_context *ctx, uint32_t rows, const uint64_t seed[8], uint64_t *out_host);
long sealhip_debug_blake2xb(void *out, size_t outlen, const void *in, size_t inlen, const void *key, size_t keylen);
/* KSwitchKeys::save (kswitchkeys.cpp:43-85 under Serialization::Save, uncompressed): writes a RelinKeys / GaloisKeys stream
   whose keys_[i] is keys[i] (NULL = unused slot, keys_dim2 = 0) with the digit words copied straight from HBM. bytes == NULL:
   only the size is reported in *written. Needs the key level's parms_id registered. The stream round-trips through
   sealhip_kswitch_key_load_stream and is what the reference's KSwitchKeys::load reads. */
long sealhip_kswitch_keys_save(sealhip_context *ctx, const sealhip_kswitch_key *const *keys, uint32_t n_slots, void *bytes,
                               size_t capacity, size_t *written);
/* is_data_valid_for (valcheck.cpp:284-317) on device-resident ciphertexts: valid[i] = 1 iff every coefficient of
   ciphertext i is below its row's prime (what an ingesting service checks before evaluating untrusted input). */
long sealhip_is_data_valid_for(sealhip_context *ctx, uint32_t k, const uint64_t *ct, uint32_t size, size_t count,
                               uint8_t *valid);

#ifdef __cplusplus
}
#endif
#endif /* SEALHIP_H */
