/*
 * oracle/sealref.h -- CPU restatement of the Gemini-SEAL hot path (TEST INFRASTRUCTURE ONLY).
 *
 * This is the parity oracle: a plain-C restatement of the reference's algorithms for the
 * RNS-NTT / BEHZ multiply / hybrid key-switch / modulus-switch path. It is NOT part of the
 * product. Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may load it.
 *
 * Parity status: PINNED by (a) the known-answer values held by the reference's own unit tests
 * (native/tests/seal/util/{ntt,rns,polyarithsmallmod,uintarithsmallmod,galois}.cpp) and
 * (b) FNV-1a-64 digests captured from the compiled reference by the survey stage
 * (SURVEY.md Appendix A.5 / B.3 / B.5), committed under tests/golden/.
 * The reference itself is unbuildable under this round's rules (needs a cmake-generated
 * config.h and a source patch), so there is no oracle/_ref.
 *
 * Every function cites the reference file:line it follows (paths relative to
 * /root/reference/native/src/seal/).
 */
#ifndef SEALREF_H
#define SEALREF_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- word arithmetic (util/uintarith.h, util/uintarithsmallmod.h) ---- */
typedef struct
{
    uint64_t value;
    uint64_t cr[3]; /* floor(2^128/value) low, high; 2^128 mod value  (modulus.cpp:85-98) */
    int bit_count;
} ref_modulus;

int ref_modulus_init(ref_modulus *m, uint64_t value);
uint64_t ref_barrett_reduce_128(uint64_t lo, uint64_t hi, const ref_modulus *m);
uint64_t ref_barrett_reduce_63(uint64_t x, const ref_modulus *m);
uint64_t ref_multiply_uint_mod(uint64_t a, uint64_t b, const ref_modulus *m);
uint64_t ref_multiply_add_uint_mod(uint64_t a, uint64_t b, uint64_t c, const ref_modulus *m);
uint64_t ref_dot_product_mod(const uint64_t *a, const uint64_t *b, size_t count, const ref_modulus *m);
uint64_t ref_exponentiate_uint_mod(uint64_t a, uint64_t e, const ref_modulus *m);
int ref_try_invert_uint_mod(uint64_t value, uint64_t modulus, uint64_t *result);
uint64_t ref_shoupify(uint64_t x, uint64_t p);

/* ---- number theory (util/numth.cpp, modulus.cpp) ---- */
int ref_is_prime(uint64_t value);
int ref_get_primes(size_t ntt_size, int bit_size, size_t count, uint64_t *out);
int ref_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out);
int ref_try_minimal_primitive_root(uint64_t degree, const ref_modulus *m, uint64_t *root);

/* ---- NTT (util/ntt.cpp, util/ntt.h) ---- */
typedef struct
{
    int logn;
    size_t n;
    ref_modulus mod;
    uint64_t root;
    uint64_t inv_degree, scaled_inv_degree, reduce_precomp;
    uint64_t *root_powers, *scaled_root_powers, *inv_root_powers, *scaled_inv_root_powers;
} ref_ntt_tables;

int ref_ntt_tables_init(ref_ntt_tables *t, int logn, uint64_t modulus);
void ref_ntt_tables_free(ref_ntt_tables *t);
/* mode: 0 = PARITY (reference's uncorrected lazy butterflies), 1 = STRICT (Harvey-corrected) */
void ref_ntt_forward_lazy(uint64_t *x, const ref_ntt_tables *t, int strict);
void ref_ntt_forward(uint64_t *x, const ref_ntt_tables *t, int strict);
void ref_ntt_inverse_lazy(uint64_t *x, const ref_ntt_tables *t);
void ref_ntt_inverse(uint64_t *x, const ref_ntt_tables *t);

/* ---- coefficient-wise polynomial arithmetic (util/polyarithsmallmod.{h,cpp}) ---- */
void ref_dyadic_product_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r);
void ref_multiply_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t scalar, const ref_modulus *m, uint64_t *r);
void ref_add_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r);
void ref_sub_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r);
void ref_negate_poly_coeffmod(const uint64_t *a, size_t n, const ref_modulus *m, uint64_t *r);
void ref_modulo_poly_coeffs_63(const uint64_t *a, size_t n, const ref_modulus *m, uint64_t *r);

/* ---- BaseConverter (util/rns.cpp:452-523) ---- */
typedef struct
{
    size_t isize, osize;
    ref_modulus *ibase, *obase;
    uint64_t *inv_punct; /* [isize]  (q^_i)^{-1} mod q_i */
    uint64_t *matrix;    /* [osize][isize]  q^_i mod p_j */
} ref_base_converter;

int ref_base_converter_init(ref_base_converter *bc, const uint64_t *ibase, size_t isize, const uint64_t *obase,
                            size_t osize);
void ref_base_converter_free(ref_base_converter *bc);
void ref_fast_convert(const ref_base_converter *bc, const uint64_t *in, uint64_t *out);
void ref_fast_convert_array(const ref_base_converter *bc, const uint64_t *in, size_t count, uint64_t *out);

/* ---- RNSTool (util/rns.cpp:539-1068) ---- */
typedef struct
{
    size_t n;
    int logn;
    size_t q_size, B_size, Bsk_size;
    ref_modulus *q;         /* [q_size] */
    ref_modulus *Bsk;       /* [Bsk_size]  B..., m_sk last */
    ref_modulus m_tilde;    /* 2^32 */
    ref_modulus m_sk, gamma, t;
    ref_ntt_tables *Bsk_ntt; /* [Bsk_size] */
    ref_base_converter q_to_Bsk, q_to_m_tilde, B_to_q, B_to_m_sk;
    uint64_t *prod_B_mod_q;       /* [q_size] */
    uint64_t *inv_prod_q_mod_Bsk; /* [Bsk_size] */
    uint64_t inv_prod_B_mod_m_sk;
    uint64_t *inv_m_tilde_mod_Bsk; /* [Bsk_size] */
    uint64_t inv_prod_q_mod_m_tilde;
    uint64_t *prod_q_mod_Bsk;    /* [Bsk_size] */
    uint64_t *inv_q_last_mod_q;  /* [q_size-1] */
} ref_rns_tool;

int ref_rns_tool_init(ref_rns_tool *rt, size_t n, const uint64_t *q, size_t q_size, uint64_t t);
void ref_rns_tool_free(ref_rns_tool *rt);
void ref_fastbconv_m_tilde(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out);
void ref_sm_mrq(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out);
void ref_fast_floor(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out);
void ref_fastbconv_sk(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out);
void ref_divide_and_round_q_last_inplace(const ref_rns_tool *rt, uint64_t *in);
void ref_divide_and_round_q_last_ntt_inplace(const ref_rns_tool *rt, uint64_t *in, const ref_ntt_tables *q_tables,
                                             int strict);

/* ---- Galois (util/galois.cpp) ---- */
uint32_t ref_galois_elt_from_step(size_t n, int step, int *ok);
void ref_galois_table_ntt(int logn, uint32_t galois_elt, uint32_t *table);
void ref_apply_galois(const uint64_t *in, int logn, uint32_t galois_elt, const ref_modulus *m, uint64_t *out);
void ref_apply_galois_ntt(const uint64_t *in, int logn, uint32_t galois_elt, uint64_t *out);

/* ---- hybrid key-switch helpers (multi_special_primes.cpp) ---- */
void ref_modup_rns(const uint64_t *src_poly, uint64_t *dst_poly, size_t n, size_t n_ct_rns, size_t n_sp_rns,
                   size_t src_bundle_index, const ref_modulus *key_mod, size_t n_key_mod);
void ref_rescale_special_rns_inplace(uint64_t *poly, int is_ckks, size_t n, size_t n_ct_rns, size_t n_sp_rns,
                                     const ref_modulus *key_mod, size_t n_key_mod, const ref_ntt_tables *key_tables,
                                     int strict);

/* ---- context + Evaluator-level drivers (context.cpp, evaluator.cpp) ---- */
#define REF_SCHEME_BFV 1
#define REF_SCHEME_CKKS 2
#define REF_MODE_PARITY 0
#define REF_MODE_STRICT 1

typedef struct
{
    int scheme;
    int logn;
    size_t n;
    size_t n_key;     /* number of key-level primes */
    size_t nsp;       /* n_special_primes */
    size_t k_first;   /* n_key - nsp */
    uint64_t t;       /* plain modulus (BFV), 0 for CKKS */
    int mode;
    ref_modulus *key_mod;       /* [n_key] */
    ref_ntt_tables *key_tables; /* [n_key] */
    ref_rns_tool **rns_tools;   /* [n_key+1], indexed by k (lazily built) */
} ref_context;

int ref_context_init(ref_context *c, int scheme, int logn, const uint64_t *key_moduli, size_t n_key, size_t nsp,
                     uint64_t t, int mode);
void ref_context_free(ref_context *c);
const ref_rns_tool *ref_context_rns_tool(ref_context *c, size_t k);

/* ciphertext layouts: size x k x N row-major uint64 (ciphertext.h:359-368) */
/* evaluator.cpp:274-445; out has (sa+sb-1) polys of k rows */
int ref_bfv_multiply(ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                     uint64_t *out);
/* evaluator.cpp:447-527 */
int ref_ckks_multiply(ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                      uint64_t *out);
/* evaluator.cpp:560-702 / :704-770: Evaluator::square as its own path (size-2 operands; other sizes fall through to multiply) */
int ref_bfv_square(ref_context *c, size_t k, const uint64_t *a, size_t sa, uint64_t *out);
int ref_ckks_square(ref_context *c, size_t k, const uint64_t *a, size_t sa, uint64_t *out);
/* evaluator.cpp:2259-2368; ct = 2 polys (k rows) updated in place; target = k rows;
   key = digits x 2 x n_key x N (K1 layout, keygenerator.cpp:325-369) */
int ref_switch_key_inplace(ref_context *c, size_t k, uint64_t *ct, const uint64_t *target, const uint64_t *key);
/* SURVEY 8(e) "latency mode": the digits of ONE key switch split across devices -- partial inner products over the digits
   [j0, j1) as canonical residues (2 x (k + nsp) x n words), summed by an all-reduce, then the rest of the key switch */
int ref_switch_key_partial(ref_context *c, size_t k, const uint64_t *target, const uint64_t *key, size_t j0, size_t j1,
                           uint64_t *partial);
int ref_switch_key_finish(ref_context *c, size_t k, uint64_t *ct, const uint64_t *partial_sum);
/* evaluator.cpp:772-827; ct has `size` polys, keys[i] is the key for RelinKeys::get_index(i+2) */
int ref_relinearize(ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *const *keys);
/* evaluator.cpp:829-892 (BFV mod_switch_to_next / CKKS rescale_to_next); out has `size` polys of k-1 rows */
int ref_mod_switch_scale_to_next(ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out);
/* evaluator.cpp:894-957 (CKKS mod_switch_to_next) */
int ref_mod_switch_drop_to_next(ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out);
/* evaluator.cpp:1841-1943; ct = 2 polys in place */
int ref_apply_galois_inplace(ref_context *c, size_t k, uint64_t *ct, uint32_t galois_elt, const uint64_t *key);

/* ---- compositions of the primitives above. The reference holds no fixture for these at
   Evaluator level that can be restated without an Encryptor (its tests encrypt first), so they are pinned only
   through their primitives (add/sub/negate/dyadic KATs, NTT KATs) and, for multiply_plain, by an independent
   schoolbook negacyclic product in tests/test_oracle.py. ---- */
/* evaluator.cpp:65-88 */
void ref_evaluator_negate(const ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out);
/* evaluator.cpp:90-151 / :174-233; out has max(sa, sb) polys */
void ref_evaluator_add(const ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                       uint64_t *out);
void ref_evaluator_sub(const ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                       uint64_t *out);
/* evaluator.cpp:1605-1646; plain_ntt = k x N, in place */
void ref_multiply_plain_ntt(const ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *plain_ntt);
/* evaluator.cpp:1475-1603, generic path with fast plain lift; plain = N coefficients < t; returns -1 when some
   q_i <= t (the multi-precision lift is not restated) */
int ref_multiply_plain(const ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *plain);
/* ciphertext.h:471-476 */
int ref_is_transparent(const ref_context *c, size_t k, const uint64_t *ct, size_t size);

/* ---- the steps either side of the path, restated for an end-to-end SEMANTIC check (encrypt ->
   evaluate -> decrypt == plaintext arithmetic). Sampling uses splitmix64 (the reference's Blake2/SHAKE PRNG is a
   client-side detail; no bit parity with its random streams is claimed or needed). ---- */
/* uniform ternary secret {-1,0,1}^n and a centred-binomial error (sigma ~ 3.2, util/rlwe.cpp:25-95 samples a
   clipped normal of the same width) */
void ref_sample_ternary(int8_t *s, size_t n, uint64_t *state);
void ref_sample_noise(int8_t *e, size_t n, uint64_t *state);
/* small signed polynomial -> RNS rows (key primes 0..rows-1), optionally NTT form */
void ref_small_poly_to_rns(const ref_context *c, const int8_t *s, size_t rows, int to_ntt, uint64_t *out);
/* util/rlwe.cpp:204-300 without seed saving: ct = ([-(a*s + e)]_q, a) over key primes 0..rows-1;
   sk_ntt = secret key in NTT form (row stride n) */
void ref_encrypt_zero_symmetric(const ref_context *c, size_t rows, const uint64_t *sk_ntt, int is_ntt_form,
                                uint64_t *state, uint64_t *ct);
/* the same with the samples handed in (a uniform in NTT form, e small signed); a_ntt may alias c1 */
void ref_encrypt_zero_symmetric_given(const ref_context *c, size_t rows, const uint64_t *sk_ntt, int is_ntt_form,
                                      const uint64_t *a_ntt, const int32_t *e, uint64_t *ct);
/* util/rlwe.cpp:140-202 with the samples handed in: pk = 2 x rows x N (NTT form), u ternary, e = 2 x N noise */
void ref_encrypt_zero_asymmetric_given(const ref_context *c, size_t rows, const uint64_t *pk, int is_ntt_form,
                                       const int32_t *u, const int32_t *e, uint64_t *ct);
/* util/scalingvariant.cpp:15-52 / :54-92: c0 (k x N) +-= round(q * plain / t); Evaluator::add_plain / sub_plain for BFV */
void ref_multiply_add_plain_with_scaling_variant(const ref_context *c, size_t k, const uint64_t *plain, int sub,
                                                 uint64_t *c0);
/* ---- BatchEncoder (batchencoder.cpp:70-154, :339-376); plain_tables = NTTTables(logn, t) ---- */
void ref_batch_index_map(int logn, uint32_t *map);
void ref_batch_encode(const ref_ntt_tables *plain_tables, const uint64_t *values, size_t count, uint64_t *plain);
void ref_batch_decode(const ref_ntt_tables *plain_tables, const uint64_t *plain, size_t count, uint64_t *values);
void ref_batch_encode_signed(const ref_ntt_tables *plain_tables, const int64_t *values, size_t count, uint64_t *plain);
void ref_batch_decode_signed(const ref_ntt_tables *plain_tables, const uint64_t *plain, size_t count, int64_t *values);
/* ---- CKKSEncoder (ckks.cpp:14-77, ckks.h:405-747), double precision ---- */
typedef struct
{
    int logn;
    size_t n;
    uint32_t *index_map; /* matrix_reps_index_map_, generator 5 */
    double *roots;       /* n complex numbers (re, im): roots_[i] = zeta^{bitrev(i)}, zeta = exp(2 pi i / 2n) */
    double *inv_roots;   /* conjugates */
} ref_ckks_encoder;
int ref_ckks_encoder_init(ref_ckks_encoder *enc, int logn);
void ref_ckks_encoder_free(ref_ckks_encoder *enc);
/* values: n_values <= n/2 complex numbers (re, im interleaved); out: rows x n in NTT form over key primes 0..rows-1.
 * 0 ok, -1 scale out of bounds, -2 encoded values are too large */
int ref_ckks_encode(const ref_context *c, const ref_ckks_encoder *enc, size_t rows, const double *values, size_t n_values,
                    double scale, uint64_t *out);
int ref_ckks_encode_value(const ref_context *c, size_t rows, double value, double scale, uint64_t *out);
int ref_ckks_encode_int64(const ref_context *c, size_t rows, int64_t value, uint64_t *out);
/* plain: rows x n (NTT form) -> n/2 complex numbers */
int ref_ckks_decode(const ref_context *c, const ref_ckks_encoder *enc, size_t rows, const uint64_t *plain, double scale,
                    double *values);
/* Encryptor::encrypt (BFV, symmetric): encrypt_zero + multiply_add_plain_with_scaling_variant (util/scalingvariant.cpp:15-52) */
void ref_bfv_encrypt_symmetric(const ref_context *c, size_t k, const uint64_t *sk_ntt, const uint64_t *plain,
                               uint64_t *state, uint64_t *ct);
/* KeyGenerator::generate_one_kswitch_key (keygenerator.cpp:325-369): key = digits x 2 x n_key x N; new_key_ntt has
   k_first rows (NTT form) */
void ref_generate_kswitch_key(const ref_context *c, const uint64_t *sk_ntt, const uint64_t *new_key_ntt,
                              uint64_t *state, uint64_t *key);
/* Decryptor::dot_product_ct_sk_array (decryptor.cpp:218-265): sk_powers = (size-1) polys (s, s^2, ...) in NTT form,
   each with key-level row stride (n_key rows); out = k rows, same form as the ciphertext */
void ref_dot_product_ct_sk(const ref_context *c, size_t k, const uint64_t *ct, size_t size, int is_ntt_form,
                           const uint64_t *sk_powers, uint64_t *out);
/* RNSTool::decrypt_scale_and_round (rns.cpp:1070-1126): k rows -> N coefficients mod t */
int ref_decrypt_scale_and_round(ref_context *c, size_t k, const uint64_t *in, uint64_t *out);

/* ---- synthetic data helpers shared by tests ---- */
uint64_t ref_splitmix64(uint64_t *state);
uint64_t ref_fnv1a64(const uint64_t *words, size_t count);
void ref_fill_rows(uint64_t *dst, size_t rows, size_t n, const uint64_t *moduli, uint64_t *state);

#ifdef __cplusplus
}
#endif
/* seeded ciphertexts -- BLAKE2Xb, BlakePRNG and sample_poly_uniform restated */
int ref_blake2xb(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen, const uint8_t *key, size_t keylen);
void ref_expand_seed(const uint64_t seed[8], const uint64_t *moduli, size_t rows, size_t n, uint64_t *dst);

#endif
