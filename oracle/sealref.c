#define _GNU_SOURCE
/*
 * oracle/sealref.c -- CPU restatement of the Gemini-SEAL hot path (TEST INFRASTRUCTURE ONLY).
 * See sealref.h for scope, parity status and the rules about who may load this file.
 * All file:line citations are relative to /root/reference/native/src/seal/.
 */
#include "sealref.h"

#include <stdlib.h>
#include <math.h>
#include <string.h>

typedef unsigned __int128 u128;

static inline uint64_t mulhi64(uint64_t a, uint64_t b)
{
    return (uint64_t)(((u128)a * b) >> 64);
}

/* ------------------------------------------------------------------------------------------
 * Modulus / Barrett constants (modulus.cpp:66-105)
 * ---------------------------------------------------------------------------------------- */
int ref_modulus_init(ref_modulus *m, uint64_t value)
{
    memset(m, 0, sizeof(*m));
    if (value == 0)
        return 0;
    if ((value >> 61) != 0 || value == 1)
        return -1;
    m->value = value;
    int bits = 0;
    for (uint64_t v = value; v; v >>= 1)
        bits++;
    m->bit_count = bits;
    /* floor(2^128 / value): divide (2^128 - 1) then fix up */
    u128 all = ~(u128)0;
    u128 q = all / value;
    u128 r = all % value;
    if (r == (u128)(value - 1))
    {
        q += 1;
        r = 0;
    }
    else
    {
        r += 1;
    }
    m->cr[0] = (uint64_t)q;
    m->cr[1] = (uint64_t)(q >> 64);
    m->cr[2] = (uint64_t)r;
    return 0;
}

/* util/uintarithsmallmod.h:140-178 */
uint64_t ref_barrett_reduce_128(uint64_t lo, uint64_t hi, const ref_modulus *m)
{
    uint64_t cr0 = m->cr[0], cr1 = m->cr[1], p = m->value;
    uint64_t carry = mulhi64(lo, cr0);
    u128 t = (u128)lo * cr1;
    uint64_t tmp1 = (uint64_t)t + carry;
    uint64_t tmp3 = (uint64_t)(t >> 64) + (tmp1 < (uint64_t)t);
    u128 u = (u128)hi * cr0;
    uint64_t tmp1b = tmp1 + (uint64_t)u;
    uint64_t carry2 = (uint64_t)(u >> 64) + (tmp1b < tmp1);
    uint64_t q = hi * cr1 + tmp3 + carry2;
    uint64_t r = lo - q * p;
    return r - (p & (uint64_t)(-(int64_t)(r >= p)));
}

/* util/uintarithsmallmod.h:181-207 */
uint64_t ref_barrett_reduce_63(uint64_t x, const ref_modulus *m)
{
    uint64_t q = mulhi64(x, m->cr[1]);
    uint64_t r = x - q * m->value;
    return r - (m->value & (uint64_t)(-(int64_t)(r >= m->value)));
}

/* util/uintarithsmallmod.h:209-221 */
uint64_t ref_multiply_uint_mod(uint64_t a, uint64_t b, const ref_modulus *m)
{
    u128 z = (u128)a * b;
    return ref_barrett_reduce_128((uint64_t)z, (uint64_t)(z >> 64), m);
}

/* util/uintarithsmallmod.h:282-290 */
uint64_t ref_multiply_add_uint_mod(uint64_t a, uint64_t b, uint64_t c, const ref_modulus *m)
{
    u128 z = (u128)a * b;
    uint64_t lo = (uint64_t)z + c;
    uint64_t hi = (uint64_t)(z >> 64) + (lo < (uint64_t)z);
    return ref_barrett_reduce_128(lo, hi, m);
}

/* util/uintarithsmallmod.cpp:110-173 (chunks of 16 from the tail; 128-bit wrapping accumulate) */
uint64_t ref_dot_product_mod(const uint64_t *a, const uint64_t *b, size_t count, const ref_modulus *m)
{
    if (count == 0)
        return 0;
    u128 acc = 0;
    size_t head = count;
    if (count > 16)
    {
        acc = ref_dot_product_mod(a + 16, b + 16, count - 16, m);
        head = 16;
    }
    for (size_t i = 0; i < head; i++)
        acc += (u128)a[i] * b[i];
    return ref_barrett_reduce_128((uint64_t)acc, (uint64_t)(acc >> 64), m);
}

/* util/uintarithsmallmod.cpp:18-64 */
uint64_t ref_exponentiate_uint_mod(uint64_t operand, uint64_t exponent, const ref_modulus *m)
{
    if (exponent == 0)
        return 1;
    if (exponent == 1)
        return operand;
    uint64_t power = operand, intermediate = 1;
    for (;;)
    {
        if (exponent & 1)
            intermediate = ref_multiply_uint_mod(power, intermediate, m);
        exponent >>= 1;
        if (!exponent)
            break;
        power = ref_multiply_uint_mod(power, power, m);
    }
    return intermediate;
}

/* util/numth.cpp:61-88 + util/numth.h:78-120 (xgcd); result normalised to [0, modulus) */
int ref_try_invert_uint_mod(uint64_t value, uint64_t modulus, uint64_t *result)
{
    if (value == 0)
        return 0;
    uint64_t x = value, y = modulus;
    __int128 prev_a = 1, a = 0;
    while (y != 0)
    {
        uint64_t q = x / y, r = x % y;
        x = y;
        y = r;
        __int128 tmp = a;
        a = prev_a - (__int128)q * a;
        prev_a = tmp;
    }
    if (x != 1)
        return 0;
    __int128 res = prev_a % (__int128)modulus;
    if (res < 0)
        res += modulus;
    *result = (uint64_t)res;
    return 1;
}

/* util/ntt.cpp:19-24: floor(x * 2^64 / p) (low word) */
uint64_t ref_shoupify(uint64_t x, uint64_t p)
{
    return (uint64_t)((((u128)x) << 64) / p);
}

/* ------------------------------------------------------------------------------------------
 * Number theory (util/numth.cpp)
 * ---------------------------------------------------------------------------------------- */
/* util/numth.cpp:179-275 uses randomised Miller-Rabin; this is the deterministic 64-bit
   variant (bases 2..37), which returns the same answer for every 64-bit input. */
int ref_is_prime(uint64_t value)
{
    if (value < 2)
        return 0;
    static const uint64_t small[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++)
    {
        if (value == small[i])
            return 1;
        if (value % small[i] == 0)
            return 0;
    }
    uint64_t d = value - 1;
    int r = 0;
    while (!(d & 1))
    {
        d >>= 1;
        r++;
    }
    for (size_t i = 0; i < sizeof(small) / sizeof(small[0]); i++)
    {
        uint64_t a = small[i] % value, x = 1, e = d, base = a;
        while (e)
        {
            if (e & 1)
                x = (uint64_t)(((u128)x * base) % value);
            base = (uint64_t)(((u128)base * base) % value);
            e >>= 1;
        }
        if (x == 1 || x == value - 1)
            continue;
        int composite = 1;
        for (int j = 1; j < r; j++)
        {
            x = (uint64_t)(((u128)x * x) % value);
            if (x == value - 1)
            {
                composite = 0;
                break;
            }
        }
        if (composite)
            return 0;
    }
    return 1;
}

/* util/numth.cpp:277-323: primes = 1 (mod 2*ntt_size) below 2^bit_size, largest first */
int ref_get_primes(size_t ntt_size, int bit_size, size_t count, uint64_t *out)
{
    uint64_t factor = 2 * (uint64_t)ntt_size;
    uint64_t value = (uint64_t)1 << bit_size;
    if (value < factor)
        return -1;
    value = value - factor + 1;
    uint64_t lower_bound = (uint64_t)1 << (bit_size - 1);
    size_t found = 0;
    while (found < count && value > lower_bound)
    {
        if (ref_is_prime(value))
            out[found++] = value;
        value -= factor;
    }
    return found == count ? 0 : -1;
}

/* modulus.cpp:134-172: per bit size take the `count` largest primes, hand them out smallest first */
int ref_coeff_modulus_create(size_t n, const int *bit_sizes, size_t count, uint64_t *out)
{
    int rc = 0;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * (count ? count : 1));
    char *done = (char *)calloc(count ? count : 1, 1);
    for (size_t i = 0; i < count && !rc; i++)
    {
        if (done[i])
            continue;
        size_t cnt = 0;
        for (size_t j = i; j < count; j++)
            if (bit_sizes[j] == bit_sizes[i])
                cnt++;
        if (ref_get_primes(n, bit_sizes[i], cnt, tmp))
        {
            rc = -1;
            break;
        }
        size_t back = cnt;
        for (size_t j = i; j < count; j++)
        {
            if (bit_sizes[j] == bit_sizes[i])
            {
                out[j] = tmp[--back];
                done[j] = 1;
            }
        }
    }
    free(tmp);
    free(done);
    return rc;
}

/* util/numth.cpp:325-424: the reference draws a random primitive root and then walks all odd
   powers to find the minimum; the result (the minimal primitive degree-th root) is deterministic. */
int ref_try_minimal_primitive_root(uint64_t degree, const ref_modulus *m, uint64_t *root_out)
{
    uint64_t p = m->value;
    uint64_t group = p - 1;
    uint64_t quot = group / degree;
    if (group - quot * degree != 0)
        return 0;
    uint64_t root = 0;
    int found = 0;
    for (uint64_t cand = 2; cand < p && cand < 100000; cand++)
    {
        uint64_t r = ref_exponentiate_uint_mod(cand, quot, m);
        if (r != 0 && ref_exponentiate_uint_mod(r, degree >> 1, m) == p - 1)
        {
            root = r;
            found = 1;
            break;
        }
    }
    if (!found)
        return 0;
    uint64_t generator_sq = ref_multiply_uint_mod(root, root, m);
    uint64_t current = root;
    for (uint64_t i = 0; i < degree; i++)
    {
        if (current < root)
            root = current;
        current = ref_multiply_uint_mod(current, generator_sq, m);
    }
    *root_out = root;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * NTT tables (util/ntt.cpp:37-119)
 * ---------------------------------------------------------------------------------------- */
static uint32_t reverse_bits32(uint32_t x, int bit_count)
{
    if (bit_count == 0)
        return 0;
    x = ((x & 0xaaaaaaaau) >> 1) | ((x & 0x55555555u) << 1);
    x = ((x & 0xccccccccu) >> 2) | ((x & 0x33333333u) << 2);
    x = ((x & 0xf0f0f0f0u) >> 4) | ((x & 0x0f0f0f0fu) << 4);
    x = ((x & 0xff00ff00u) >> 8) | ((x & 0x00ff00ffu) << 8);
    x = (x >> 16) | (x << 16);
    return x >> (32 - bit_count);
}

/* util/ntt.cpp:101-111 */
static void powers_of_root_bitrev(uint64_t root, int logn, const ref_modulus *m, uint64_t *dst)
{
    size_t n = (size_t)1 << logn;
    dst[0] = 1;
    uint64_t prev = 1;
    for (size_t i = 1; i < n; i++)
    {
        prev = ref_multiply_uint_mod(prev, root, m);
        dst[reverse_bits32((uint32_t)i, logn)] = prev;
    }
}

int ref_ntt_tables_init(ref_ntt_tables *t, int logn, uint64_t modulus)
{
    memset(t, 0, sizeof(*t));
    t->logn = logn;
    t->n = (size_t)1 << logn;
    size_t n = t->n;
    if (ref_modulus_init(&t->mod, modulus))
        return -1;
    if (!ref_try_minimal_primitive_root(2 * (uint64_t)n, &t->mod, &t->root))
        return -1;
    uint64_t inverse_root;
    if (!ref_try_invert_uint_mod(t->root, modulus, &inverse_root))
        return -1;
    if (!ref_try_invert_uint_mod((uint64_t)n, modulus, &t->inv_degree))
        return -1;
    t->scaled_inv_degree = ref_shoupify(t->inv_degree, modulus);
    t->reduce_precomp = ref_shoupify(1, modulus);
    t->root_powers = (uint64_t *)malloc(4 * n * sizeof(uint64_t));
    t->scaled_root_powers = t->root_powers + n;
    t->inv_root_powers = t->root_powers + 2 * n;
    t->scaled_inv_root_powers = t->root_powers + 3 * n;
    powers_of_root_bitrev(t->root, logn, &t->mod, t->root_powers);
    for (size_t i = 0; i < n; i++)
        t->scaled_root_powers[i] = ref_shoupify(t->root_powers[i], modulus);
    powers_of_root_bitrev(inverse_root, logn, &t->mod, t->inv_root_powers);
    /* ntt.cpp:85-95: re-order so the inverse transform reads sequentially */
    uint64_t *temp = (uint64_t *)malloc(n * sizeof(uint64_t));
    uint64_t *tp = temp + 1;
    for (size_t m = n >> 1; m > 0; m >>= 1)
        for (size_t i = 0; i < m; i++)
            *tp++ = t->inv_root_powers[m + i];
    memcpy(t->inv_root_powers + 1, temp + 1, (n - 1) * sizeof(uint64_t));
    free(temp);
    /* ntt.cpp:97: merge n^{-1} into the last entry */
    t->inv_root_powers[n - 1] = ref_multiply_uint_mod(t->inv_root_powers[n - 1], t->inv_degree, &t->mod);
    for (size_t i = 0; i < n; i++)
        t->scaled_inv_root_powers[i] = ref_shoupify(t->inv_root_powers[i], modulus);
    return 0;
}

void ref_ntt_tables_free(ref_ntt_tables *t)
{
    free(t->root_powers);
    memset(t, 0, sizeof(*t));
}

/* ------------------------------------------------------------------------------------------
 * NTT transforms (util/ntt.cpp:213-404, util/ntt.h:225-334)
 * ---------------------------------------------------------------------------------------- */
static inline uint64_t mulmod_lazy(uint64_t x, uint64_t y, uint64_t yshoup, uint64_t p)
{
    uint64_t q = mulhi64(x, yshoup); /* ntt.cpp:230-234 */
    return x * y - q * p;
}

/* ntt.cpp:292-342. The hand-unrolled loops of the reference are scheduling only; the dataflow
   is: layer with gap h (h = n/2 .. 1), m = n/(2h) groups, group r uses table entry (m + r),
   consumed sequentially. PARITY: no correction of u (ForwardLazy, :245-252), last layer applies
   reduceBarrettLazy to u (ForwardLazyLast, :254-261). STRICT: Harvey's u -= 2p if u >= 2p. */
void ref_ntt_forward_lazy(uint64_t *x, const ref_ntt_tables *t, int strict)
{
    const uint64_t p = t->mod.value, Lp = p << 1, rdp = t->reduce_precomp;
    const size_t n = t->n;
    const uint64_t *w = t->root_powers + 1;
    const uint64_t *ws = t->scaled_root_powers + 1;
    for (size_t m = 1, h = n >> 1; h >= 1; m <<= 1, h >>= 1)
    {
        for (size_t r = 0; r < m; r++, w++, ws++)
        {
            uint64_t W = *w, Ws = *ws;
            uint64_t *x0 = x + 2 * h * r, *x1 = x0 + h;
            for (size_t i = 0; i < h; i++)
            {
                uint64_t u = x0[i];
                if (strict)
                    u -= (u >= Lp) ? Lp : 0;
                else if (h == 1)
                    u = u - mulhi64(u, rdp) * p; /* reduceBarrettLazy, ntt.cpp:237-241 */
                uint64_t v = mulmod_lazy(x1[i], W, Ws, p);
                x0[i] = u + v;
                x1[i] = u - v + Lp;
            }
        }
    }
}

/* ntt.h:225-246 */
void ref_ntt_forward(uint64_t *x, const ref_ntt_tables *t, int strict)
{
    ref_ntt_forward_lazy(x, t, strict);
    const uint64_t p = t->mod.value, two_p = p * 2;
    for (size_t i = 0; i < t->n; i++)
    {
        if (x[i] >= two_p)
            x[i] -= two_p;
        if (x[i] >= p)
            x[i] -= p;
    }
}

/* ntt.cpp:345-404 (valid for n >= 8 exactly like the reference; for n < 8 the reference reads
   past its tables, so the oracle refuses those sizes in the callers). */
void ref_ntt_inverse_lazy(uint64_t *x, const ref_ntt_tables *t)
{
    const uint64_t p = t->mod.value, Lp = p << 1;
    const size_t n = t->n;
    const uint64_t *w = t->inv_root_powers + 1;
    const uint64_t *ws = t->scaled_inv_root_powers + 1;
    for (size_t h = 1, m = n >> 1; m > 1; h <<= 1, m >>= 1)
    {
        for (size_t r = 0; r < m; r++, w++, ws++)
        {
            uint64_t W = *w, Ws = *ws;
            uint64_t *x0 = x + 2 * h * r, *x1 = x0 + h;
            for (size_t i = 0; i < h; i++)
            {
                uint64_t u = x0[i], v = x1[i];
                uint64_t tt = u + v;
                tt -= (tt >= Lp) ? Lp : 0; /* select(Lp, t < Lp), ntt.cpp:269 */
                x0[i] = tt;
                x1[i] = mulmod_lazy(u - v + Lp, W, Ws, p);
            }
        }
    }
    /* last layer, n^{-1} merged (ntt.cpp:393-402) */
    const uint64_t inv_n = t->inv_degree, inv_n_s = t->scaled_inv_degree;
    const uint64_t W = *w, Ws = *ws;
    uint64_t *x0 = x, *x1 = x + n / 2;
    for (size_t i = 0; i < n / 2; i++)
    {
        uint64_t u = x0[i], v = x1[i];
        uint64_t tt = u + v;
        tt -= (tt >= Lp) ? Lp : 0;
        x0[i] = mulmod_lazy(tt, inv_n, inv_n_s, p);
        x1[i] = mulmod_lazy(u - v + Lp, W, Ws, p);
    }
}

/* ntt.h:318-334 */
void ref_ntt_inverse(uint64_t *x, const ref_ntt_tables *t)
{
    ref_ntt_inverse_lazy(x, t);
    const uint64_t p = t->mod.value;
    for (size_t i = 0; i < t->n; i++)
        if (x[i] >= p)
            x[i] -= p;
}

/* ------------------------------------------------------------------------------------------
 * Coefficient-wise arithmetic (util/polyarithsmallmod.{h,cpp})
 * ---------------------------------------------------------------------------------------- */
/* polyarithsmallmod.cpp:63-117 */
void ref_dyadic_product_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r)
{
    for (size_t i = 0; i < n; i++)
        r[i] = ref_multiply_uint_mod(a[i], b[i], m);
}

/* polyarithsmallmod.cpp:15-61 */
void ref_multiply_poly_scalar_coeffmod(const uint64_t *a, size_t n, uint64_t scalar, const ref_modulus *m,
                                       uint64_t *r)
{
    for (size_t i = 0; i < n; i++)
        r[i] = ref_multiply_uint_mod(a[i], scalar, m);
}

/* polyarithsmallmod.h:261-299 */
void ref_add_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r)
{
    const uint64_t p = m->value;
    for (size_t i = 0; i < n; i++)
    {
        uint64_t s = a[i] + b[i];
        r[i] = s - (p & (uint64_t)(-(int64_t)(s >= p)));
    }
}

/* polyarithsmallmod.h:366-404 */
void ref_sub_poly_coeffmod(const uint64_t *a, const uint64_t *b, size_t n, const ref_modulus *m, uint64_t *r)
{
    const uint64_t p = m->value;
    for (size_t i = 0; i < n; i++)
    {
        uint64_t d = a[i] - b[i];
        uint64_t borrow = a[i] < b[i];
        r[i] = d + (p & (uint64_t)(-(int64_t)borrow));
    }
}

/* polyarithsmallmod.h:176-205 */
void ref_negate_poly_coeffmod(const uint64_t *a, size_t n, const ref_modulus *m, uint64_t *r)
{
    const uint64_t p = m->value;
    for (size_t i = 0; i < n; i++)
        r[i] = (p - a[i]) & (uint64_t)(-(int64_t)(a[i] != 0));
}

/* polyarithsmallmod.h:98-120 */
void ref_modulo_poly_coeffs_63(const uint64_t *a, size_t n, const ref_modulus *m, uint64_t *r)
{
    for (size_t i = 0; i < n; i++)
        r[i] = ref_barrett_reduce_63(a[i], m);
}

static inline uint64_t sub_uint64_mod(uint64_t a, uint64_t b, uint64_t p)
{
    uint64_t d = a - b; /* uintarithsmallmod.h:116-137 */
    return d + (p & (uint64_t)(-(int64_t)(a < b)));
}

static inline uint64_t negate_uint_mod(uint64_t a, uint64_t p)
{
    return (p - a) & (uint64_t)(-(int64_t)(a != 0)); /* uintarithsmallmod.h:51-65 */
}

/* ------------------------------------------------------------------------------------------
 * BaseConverter (util/rns.cpp:237-290, 452-523)
 * ---------------------------------------------------------------------------------------- */
int ref_base_converter_init(ref_base_converter *bc, const uint64_t *ibase, size_t isize, const uint64_t *obase,
                            size_t osize)
{
    memset(bc, 0, sizeof(*bc));
    bc->isize = isize;
    bc->osize = osize;
    bc->ibase = (ref_modulus *)calloc(isize, sizeof(ref_modulus));
    bc->obase = (ref_modulus *)calloc(osize, sizeof(ref_modulus));
    bc->inv_punct = (uint64_t *)calloc(isize, sizeof(uint64_t));
    bc->matrix = (uint64_t *)calloc(isize * osize, sizeof(uint64_t));
    for (size_t i = 0; i < isize; i++)
        if (ref_modulus_init(&bc->ibase[i], ibase[i]))
            return -1;
    for (size_t j = 0; j < osize; j++)
        if (ref_modulus_init(&bc->obase[j], obase[j]))
            return -1;
    /* rns.cpp:269-286: inverse of the punctured product modulo its own prime */
    for (size_t i = 0; i < isize; i++)
    {
        uint64_t prod = 1 % ibase[i];
        for (size_t l = 0; l < isize; l++)
            if (l != i)
                prod = ref_multiply_uint_mod(prod, ibase[l] % ibase[i], &bc->ibase[i]);
        if (isize == 1)
            bc->inv_punct[i] = 1;
        else if (!ref_try_invert_uint_mod(prod, ibase[i], &bc->inv_punct[i]))
            return -1;
    }
    /* rns.cpp:512-522: base_change_matrix[j][i] = (prod_{l != i} q_l) mod p_j */
    for (size_t j = 0; j < osize; j++)
        for (size_t i = 0; i < isize; i++)
        {
            uint64_t prod = 1 % obase[j];
            for (size_t l = 0; l < isize; l++)
                if (l != i)
                    prod = ref_multiply_uint_mod(prod, ibase[l] % obase[j], &bc->obase[j]);
            bc->matrix[j * isize + i] = prod;
        }
    return 0;
}

void ref_base_converter_free(ref_base_converter *bc)
{
    free(bc->ibase);
    free(bc->obase);
    free(bc->inv_punct);
    free(bc->matrix);
    memset(bc, 0, sizeof(*bc));
}

/* rns.cpp:452-467 */
void ref_fast_convert(const ref_base_converter *bc, const uint64_t *in, uint64_t *out)
{
    uint64_t temp[64];
    for (size_t i = 0; i < bc->isize; i++)
        temp[i] = ref_multiply_uint_mod(in[i], bc->inv_punct[i], &bc->ibase[i]);
    for (size_t j = 0; j < bc->osize; j++)
        out[j] = ref_dot_product_mod(temp, bc->matrix + j * bc->isize, bc->isize, &bc->obase[j]);
}

/* rns.cpp:469-496 (in: isize x count row-major; out: osize x count) */
void ref_fast_convert_array(const ref_base_converter *bc, const uint64_t *in, size_t count, uint64_t *out)
{
    size_t is = bc->isize;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * count * is);
    for (size_t i = 0; i < is; i++)
        for (size_t k = 0; k < count; k++)
            temp[k * is + i] = ref_multiply_uint_mod(in[i * count + k], bc->inv_punct[i], &bc->ibase[i]);
    for (size_t j = 0; j < bc->osize; j++)
        for (size_t k = 0; k < count; k++)
            out[j * count + k] = ref_dot_product_mod(temp + k * is, bc->matrix + j * is, is, &bc->obase[j]);
    free(temp);
}

/* ------------------------------------------------------------------------------------------
 * RNSTool (util/rns.cpp:539-729)
 * ---------------------------------------------------------------------------------------- */
static uint64_t prod_mod(const uint64_t *vals, size_t count, const ref_modulus *m)
{
    uint64_t prod = 1 % m->value;
    for (size_t i = 0; i < count; i++)
        prod = ref_multiply_uint_mod(prod, vals[i] % m->value, m);
    return prod;
}

/* bit length of the product of the base (rns.cpp:566 get_significant_bit_count_uint(base_prod)) */
static int product_bit_count(const uint64_t *vals, size_t count)
{
    uint64_t acc[66];
    memset(acc, 0, sizeof(acc));
    acc[0] = 1;
    size_t len = 1;
    for (size_t i = 0; i < count; i++)
    {
        uint64_t carry = 0;
        for (size_t j = 0; j < len; j++)
        {
            u128 z = (u128)acc[j] * vals[i] + carry;
            acc[j] = (uint64_t)z;
            carry = (uint64_t)(z >> 64);
        }
        if (carry)
            acc[len++] = carry;
    }
    int bits = 0;
    for (uint64_t v = acc[len - 1]; v; v >>= 1)
        bits++;
    return bits + 64 * (int)(len - 1);
}

int ref_rns_tool_init(ref_rns_tool *rt, size_t n, const uint64_t *q, size_t q_size, uint64_t t)
{
    memset(rt, 0, sizeof(*rt));
    rt->n = n;
    int logn = 0;
    while (((size_t)1 << logn) < n)
        logn++;
    rt->logn = logn;
    rt->q_size = q_size;
    ref_modulus_init(&rt->t, t);
    int total_bits = product_bit_count(q, q_size);
    size_t B_size = q_size;
    /* rns.cpp:568-573 */
    if (32 + rt->t.bit_count + total_bits >= 61 * (int)q_size + 61)
        B_size++;
    rt->B_size = B_size;
    rt->Bsk_size = B_size + 1;
    size_t n_aux = rt->Bsk_size + 1; /* base_Bsk_m_tilde_size primes sampled: m_sk, gamma, B... */
    uint64_t *aux = (uint64_t *)malloc(sizeof(uint64_t) * n_aux);
    /* rns.cpp:587: SEAL_USER_MOD_BIT_COUNT_MAX + 1 = 60-bit primes */
    if (ref_get_primes(n, 60, n_aux, aux))
    {
        free(aux);
        return -1;
    }
    ref_modulus_init(&rt->m_sk, aux[0]);
    ref_modulus_init(&rt->gamma, aux[1]);
    ref_modulus_init(&rt->m_tilde, (uint64_t)1 << 32);
    rt->q = (ref_modulus *)calloc(q_size, sizeof(ref_modulus));
    for (size_t i = 0; i < q_size; i++)
        ref_modulus_init(&rt->q[i], q[i]);
    uint64_t *Bvals = (uint64_t *)malloc(sizeof(uint64_t) * (rt->Bsk_size));
    for (size_t i = 0; i < B_size; i++)
        Bvals[i] = aux[2 + i];
    Bvals[B_size] = aux[0]; /* Bsk = B U {m_sk}, m_sk last (rns.cpp:600) */
    rt->Bsk = (ref_modulus *)calloc(rt->Bsk_size, sizeof(ref_modulus));
    for (size_t i = 0; i < rt->Bsk_size; i++)
        ref_modulus_init(&rt->Bsk[i], Bvals[i]);
    /* rns.cpp:610-620 */
    rt->Bsk_ntt = (ref_ntt_tables *)calloc(rt->Bsk_size, sizeof(ref_ntt_tables));
    for (size_t i = 0; i < rt->Bsk_size; i++)
        if (ref_ntt_tables_init(&rt->Bsk_ntt[i], logn, Bvals[i]))
            return -1;
    uint64_t mt = rt->m_tilde.value, msk = rt->m_sk.value;
    if (ref_base_converter_init(&rt->q_to_Bsk, q, q_size, Bvals, rt->Bsk_size) ||
        ref_base_converter_init(&rt->q_to_m_tilde, q, q_size, &mt, 1) ||
        ref_base_converter_init(&rt->B_to_q, Bvals, B_size, q, q_size) ||
        ref_base_converter_init(&rt->B_to_m_sk, Bvals, B_size, &msk, 1))
        return -1;
    /* rns.cpp:640-688 */
    rt->prod_B_mod_q = (uint64_t *)calloc(q_size, sizeof(uint64_t));
    for (size_t i = 0; i < q_size; i++)
        rt->prod_B_mod_q[i] = prod_mod(Bvals, B_size, &rt->q[i]);
    rt->inv_prod_q_mod_Bsk = (uint64_t *)calloc(rt->Bsk_size, sizeof(uint64_t));
    rt->prod_q_mod_Bsk = (uint64_t *)calloc(rt->Bsk_size, sizeof(uint64_t));
    rt->inv_m_tilde_mod_Bsk = (uint64_t *)calloc(rt->Bsk_size, sizeof(uint64_t));
    for (size_t i = 0; i < rt->Bsk_size; i++)
    {
        rt->prod_q_mod_Bsk[i] = prod_mod(q, q_size, &rt->Bsk[i]);
        if (!ref_try_invert_uint_mod(rt->prod_q_mod_Bsk[i], Bvals[i], &rt->inv_prod_q_mod_Bsk[i]))
            return -1;
        if (!ref_try_invert_uint_mod(mt % Bvals[i], Bvals[i], &rt->inv_m_tilde_mod_Bsk[i]))
            return -1;
    }
    uint64_t pb = prod_mod(Bvals, B_size, &rt->m_sk);
    if (!ref_try_invert_uint_mod(pb, msk, &rt->inv_prod_B_mod_m_sk))
        return -1;
    uint64_t pq = prod_mod(q, q_size, &rt->m_tilde);
    if (!ref_try_invert_uint_mod(pq, mt, &rt->inv_prod_q_mod_m_tilde))
        return -1;
    /* rns.cpp:719-728 */
    if (q_size > 1)
    {
        rt->inv_q_last_mod_q = (uint64_t *)calloc(q_size - 1, sizeof(uint64_t));
        for (size_t i = 0; i + 1 < q_size; i++)
            if (!ref_try_invert_uint_mod(q[q_size - 1], q[i], &rt->inv_q_last_mod_q[i]))
                return -1;
    }
    free(aux);
    free(Bvals);
    return 0;
}

void ref_rns_tool_free(ref_rns_tool *rt)
{
    if (rt->Bsk_ntt)
        for (size_t i = 0; i < rt->Bsk_size; i++)
            ref_ntt_tables_free(&rt->Bsk_ntt[i]);
    free(rt->Bsk_ntt);
    free(rt->q);
    free(rt->Bsk);
    ref_base_converter_free(&rt->q_to_Bsk);
    ref_base_converter_free(&rt->q_to_m_tilde);
    ref_base_converter_free(&rt->B_to_q);
    ref_base_converter_free(&rt->B_to_m_sk);
    free(rt->prod_B_mod_q);
    free(rt->inv_prod_q_mod_Bsk);
    free(rt->prod_q_mod_Bsk);
    free(rt->inv_m_tilde_mod_Bsk);
    free(rt->inv_q_last_mod_q);
    memset(rt, 0, sizeof(*rt));
}

/* rns.cpp:1025-1068: in q (q_size x n) -> out Bsk U {m_tilde} ((Bsk_size+1) x n) */
void ref_fastbconv_m_tilde(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out)
{
    size_t n = rt->n;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * rt->q_size);
    for (size_t i = 0; i < rt->q_size; i++)
        ref_multiply_poly_scalar_coeffmod(in + i * n, n, rt->m_tilde.value, &rt->q[i], temp + i * n);
    ref_fast_convert_array(&rt->q_to_Bsk, temp, n, out);
    ref_fast_convert_array(&rt->q_to_m_tilde, temp, n, out + rt->Bsk_size * n);
    free(temp);
}

/* rns.cpp:925-981: in Bsk U {m_tilde} -> out Bsk */
void ref_sm_mrq(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out)
{
    size_t n = rt->n;
    const uint64_t *in_mt = in + n * rt->Bsk_size;
    const uint64_t mt = rt->m_tilde.value, mt_div_2 = mt >> 1;
    uint64_t *r_mt = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < n; i++)
    {
        uint64_t temp = ref_multiply_uint_mod(in_mt[i], rt->inv_prod_q_mod_m_tilde, &rt->m_tilde);
        r_mt[i] = negate_uint_mod(temp, mt);
    }
    for (size_t k = 0; k < rt->Bsk_size; k++)
    {
        const ref_modulus *b = &rt->Bsk[k];
        for (size_t i = 0; i < n; i++)
        {
            uint64_t temp = r_mt[i];
            if (temp >= mt_div_2)
                temp += b->value - mt;
            out[k * n + i] = ref_multiply_uint_mod(
                ref_multiply_add_uint_mod(rt->prod_q_mod_Bsk[k], temp, in[k * n + i], b), rt->inv_m_tilde_mod_Bsk[k],
                b);
        }
    }
    free(r_mt);
}

/* rns.cpp:983-1023: in q U Bsk -> out Bsk */
void ref_fast_floor(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out)
{
    size_t n = rt->n;
    ref_fast_convert_array(&rt->q_to_Bsk, in, n, out);
    in += rt->q_size * n;
    for (size_t i = 0; i < rt->Bsk_size; i++)
    {
        const ref_modulus *b = &rt->Bsk[i];
        for (size_t k = 0; k < n; k++)
            out[i * n + k] =
                ref_multiply_uint_mod(in[i * n + k] + (b->value - out[i * n + k]), rt->inv_prod_q_mod_Bsk[i], b);
    }
}

/* rns.cpp:853-923: in Bsk -> out q (Shenoy-Kumaresan) */
void ref_fastbconv_sk(const ref_rns_tool *rt, const uint64_t *in, uint64_t *out)
{
    size_t n = rt->n;
    ref_fast_convert_array(&rt->B_to_q, in, n, out);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    uint64_t *alpha = (uint64_t *)malloc(sizeof(uint64_t) * n);
    ref_fast_convert_array(&rt->B_to_m_sk, in, n, temp);
    const uint64_t *in_sk = in + rt->B_size * n;
    const uint64_t msk = rt->m_sk.value, msk_div_2 = msk >> 1;
    for (size_t i = 0; i < n; i++)
        alpha[i] = ref_multiply_uint_mod(temp[i] + (msk - in_sk[i]), rt->inv_prod_B_mod_m_sk, &rt->m_sk);
    for (size_t i = 0; i < rt->q_size; i++)
    {
        const ref_modulus *qi = &rt->q[i];
        uint64_t pB = rt->prod_B_mod_q[i];
        for (size_t k = 0; k < n; k++)
        {
            if (alpha[k] > msk_div_2)
                out[i * n + k] = ref_multiply_add_uint_mod(pB, msk - alpha[k], out[i * n + k], qi);
            else
                out[i * n + k] = ref_multiply_add_uint_mod(qi->value - pB, alpha[k], out[i * n + k], qi);
        }
    }
    free(temp);
    free(alpha);
}

/* rns.cpp:731-775 */
void ref_divide_and_round_q_last_inplace(const ref_rns_tool *rt, uint64_t *in)
{
    size_t n = rt->n, k = rt->q_size;
    uint64_t *last = in + (k - 1) * n;
    const ref_modulus *ql = &rt->q[k - 1];
    uint64_t half = ql->value >> 1;
    for (size_t j = 0; j < n; j++)
        last[j] = ref_barrett_reduce_63(last[j] + half, ql);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i + 1 < k; i++)
    {
        const ref_modulus *qi = &rt->q[i];
        ref_modulo_poly_coeffs_63(last, n, qi, temp);
        uint64_t half_mod = ref_barrett_reduce_63(half, qi);
        for (size_t j = 0; j < n; j++)
            temp[j] = sub_uint64_mod(temp[j], half_mod, qi->value);
        ref_sub_poly_coeffmod(in + i * n, temp, n, qi, in + i * n);
        ref_multiply_poly_scalar_coeffmod(in + i * n, n, rt->inv_q_last_mod_q[i], qi, in + i * n);
    }
    free(temp);
}

/* rns.cpp:777-851 (SEAL_USER_MOD_BIT_COUNT_MAX = 59 <= 60 branch: qi_lazy = 4*qi) */
void ref_divide_and_round_q_last_ntt_inplace(const ref_rns_tool *rt, uint64_t *in, const ref_ntt_tables *q_tables,
                                             int strict)
{
    size_t n = rt->n, k = rt->q_size;
    uint64_t *last = in + (k - 1) * n;
    const ref_modulus *ql = &rt->q[k - 1];
    ref_ntt_inverse(last, &q_tables[k - 1]);
    uint64_t half = ql->value >> 1;
    for (size_t j = 0; j < n; j++)
        last[j] = ref_barrett_reduce_63(last[j] + half, ql);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i + 1 < k; i++)
    {
        const ref_modulus *qi = &rt->q[i];
        if (qi->value < ql->value)
            ref_modulo_poly_coeffs_63(last, n, qi, temp);
        else
            memcpy(temp, last, n * sizeof(uint64_t));
        uint64_t neg_half_mod = qi->value - ref_barrett_reduce_63(half, qi);
        for (size_t j = 0; j < n; j++)
            temp[j] += neg_half_mod;
        uint64_t qi_lazy = qi->value << 2;
        ref_ntt_forward_lazy(temp, &q_tables[i], strict);
        uint64_t *row = in + i * n;
        for (size_t j = 0; j < n; j++)
            row[j] = row[j] + qi_lazy - temp[j];
        ref_multiply_poly_scalar_coeffmod(row, n, rt->inv_q_last_mod_q[i], qi, row);
    }
    free(temp);
}

/* ------------------------------------------------------------------------------------------
 * Galois (util/galois.cpp)
 * ---------------------------------------------------------------------------------------- */
/* galois.cpp:49-91 (generator_ = 5, util/galois.h:169) */
uint32_t ref_galois_elt_from_step(size_t n_, int step, int *ok)
{
    uint32_t n = (uint32_t)n_;
    uint32_t m32 = n * 2;
    uint64_t m = m32;
    if (ok)
        *ok = 1;
    if (step == 0)
        return (uint32_t)(m - 1);
    int sign = step < 0;
    uint32_t pos_step = (uint32_t)(step < 0 ? -step : step);
    if (pos_step >= (n >> 1))
    {
        if (ok)
            *ok = 0;
        return 0;
    }
    pos_step &= m32 - 1;
    int s = sign ? (int)(n >> 1) - (int)pos_step : (int)pos_step;
    uint64_t gen = 5, elt = 1;
    while (s--)
    {
        elt *= gen;
        elt &= m - 1;
    }
    return (uint32_t)elt;
}

/* galois.cpp:18-47 */
void ref_galois_table_ntt(int logn, uint32_t galois_elt, uint32_t *table)
{
    size_t n = (size_t)1 << logn;
    uint32_t nm1 = (uint32_t)n - 1;
    for (size_t i = n; i < (n << 1); i++)
    {
        uint32_t reversed = reverse_bits32((uint32_t)i, logn + 1);
        uint64_t index_raw = ((uint64_t)galois_elt * (uint64_t)reversed) >> 1;
        index_raw &= (uint64_t)nm1;
        *table++ = reverse_bits32((uint32_t)index_raw, logn);
    }
}

/* galois.cpp:144-186 */
void ref_apply_galois(const uint64_t *in, int logn, uint32_t galois_elt, const ref_modulus *m, uint64_t *out)
{
    const uint64_t p = m->value;
    const uint64_t nm1 = ((uint64_t)1 << logn) - 1;
    uint64_t index_raw = 0;
    for (uint64_t i = 0; i <= nm1; i++, index_raw += galois_elt)
    {
        uint64_t index = index_raw & nm1;
        uint64_t v = in[i];
        if ((index_raw >> logn) & 1)
            v = (p - v) & (uint64_t)(-(int64_t)(v != 0));
        out[index] = v;
    }
}

/* galois.cpp:188-214 */
void ref_apply_galois_ntt(const uint64_t *in, int logn, uint32_t galois_elt, uint64_t *out)
{
    size_t n = (size_t)1 << logn;
    uint32_t *table = (uint32_t *)malloc(sizeof(uint32_t) * n);
    ref_galois_table_ntt(logn, galois_elt, table);
    for (size_t i = 0; i < n; i++)
        out[i] = in[table[i]];
    free(table);
}

/* ------------------------------------------------------------------------------------------
 * Hybrid key-switch helpers (multi_special_primes.cpp)
 * ---------------------------------------------------------------------------------------- */
/* multi_special_primes.cpp:13-19 */
static inline uint64_t mulmod_shoup(uint64_t x, uint64_t cnst, uint64_t cnst_shoup, uint64_t p)
{
    uint64_t q = mulhi64(x, cnst_shoup) * p;
    uint64_t t = x * cnst - q;
    return t - ((p & (uint64_t)(-(int64_t)(t < p))) ^ p);
}

/* multi_special_primes.cpp:80-148 */
static void modup_to_single_rns(const uint64_t *in_poly, uint64_t *dst_poly, size_t n, const size_t *idx,
                                size_t n_idx, size_t dst_idx, const ref_modulus *key_mod)
{
    if (n_idx == 1)
    {
        if (key_mod[idx[0]].value <= key_mod[dst_idx].value)
            memcpy(dst_poly, in_poly, sizeof(uint64_t) * n);
        else
            for (size_t d = 0; d < n; d++)
                dst_poly[d] = ref_barrett_reduce_63(in_poly[d], &key_mod[dst_idx]);
        return;
    }
    uint64_t inv_punch[64], punch[64];
    for (size_t a = 0; a < n_idx; a++)
    {
        uint64_t inv_prod = 1, prod = 1;
        for (size_t b = 0; b < n_idx; b++)
        {
            if (idx[a] == idx[b])
                continue;
            prod = ref_multiply_uint_mod(prod, key_mod[idx[b]].value, &key_mod[dst_idx]);
            inv_prod = ref_multiply_uint_mod(inv_prod, key_mod[idx[b]].value, &key_mod[idx[a]]);
        }
        punch[a] = prod;
        ref_try_invert_uint_mod(inv_prod, key_mod[idx[a]].value, &inv_punch[a]);
    }
    u128 *accum = (u128 *)calloc(n, sizeof(u128));
    for (size_t a = 0; a < n_idx; a++)
    {
        uint64_t p = key_mod[idx[a]].value;
        uint64_t c = inv_punch[a], cs = ref_shoupify(c, p);
        const uint64_t *src = in_poly + a * n;
        for (size_t d = 0; d < n; d++)
            accum[d] += (u128)mulmod_shoup(src[d], c, cs, p) * punch[a];
    }
    for (size_t d = 0; d < n; d++)
        dst_poly[d] = ref_barrett_reduce_128((uint64_t)accum[d], (uint64_t)(accum[d] >> 64), &key_mod[dst_idx]);
    free(accum);
}

/* multi_special_primes.cpp:151-185 */
void ref_modup_rns(const uint64_t *src_poly, uint64_t *dst_poly, size_t n, size_t n_ct_rns, size_t n_sp_rns,
                   size_t src_bundle_index, const ref_modulus *key_mod, size_t n_key_mod)
{
    size_t n_bundles = (n_ct_rns + n_sp_rns - 1) / n_sp_rns;
    size_t rns0 = src_bundle_index * n_sp_rns;
    size_t rns1 = rns0 + n_sp_rns < n_ct_rns ? rns0 + n_sp_rns : n_ct_rns;
    size_t idx[64];
    for (size_t i = rns0; i < rns1; i++)
        idx[i - rns0] = i;
    for (size_t b = 0; b < n_bundles; b++)
    {
        if (b == src_bundle_index)
            continue;
        size_t d0 = b * n_sp_rns;
        size_t d1 = d0 + n_sp_rns < n_ct_rns ? d0 + n_sp_rns : n_ct_rns;
        for (size_t d = d0; d < d1; d++)
            modup_to_single_rns(src_poly, dst_poly + d * n, n, idx, rns1 - rns0, d, key_mod);
    }
    size_t sp0 = n_key_mod - n_sp_rns;
    for (size_t k = 0; k < n_sp_rns; k++)
        modup_to_single_rns(src_poly, dst_poly + (n_ct_rns + k) * n, n, idx, rns1 - rns0, sp0 + k, key_mod);
}

/* multi_special_primes.cpp:237-304 */
void ref_rescale_special_rns_inplace(uint64_t *poly, int is_ckks, size_t n, size_t n_ct_rns, size_t n_sp_rns,
                                     const ref_modulus *key_mod, size_t n_key_mod, const ref_ntt_tables *key_tables,
                                     int strict)
{
    size_t sp0 = n_key_mod - n_sp_rns;
    uint64_t inv_hat[64];
    /* :209-234 */
    for (size_t i = 0; i < n_sp_rns; i++)
    {
        uint64_t prod = 1;
        for (size_t j = 0; j < n_sp_rns; j++)
            if (i != j)
                prod = ref_multiply_uint_mod(prod, key_mod[sp0 + j].value, &key_mod[sp0 + i]);
        ref_try_invert_uint_mod(prod, key_mod[sp0 + i].value, &inv_hat[i]);
    }
    u128 *lazy = (u128 *)malloc(sizeof(u128) * n);
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t i = 0; i < n_ct_rns; i++)
    {
        const ref_modulus *qi = &key_mod[i];
        memset(lazy, 0, sizeof(u128) * n);
        for (size_t j = 0; j < n_sp_rns; j++)
        {
            const uint64_t *ct = poly + (n_ct_rns + j) * n;
            if (n_sp_rns > 1)
            {
                /* :186-207 neg puncture product of special primes mod q_i */
                uint64_t prod = 1;
                for (size_t l = 0; l < n_sp_rns; l++)
                    if (l != j)
                        prod = ref_multiply_uint_mod(prod, key_mod[sp0 + l].value, qi);
                uint64_t neg = negate_uint_mod(prod, qi->value);
                uint64_t pj = key_mod[sp0 + j].value, c = inv_hat[j], cs = ref_shoupify(c, pj);
                for (size_t l = 0; l < n; l++)
                    lazy[l] += (u128)mulmod_shoup(ct[l], c, cs, pj) * neg;
            }
            else
            {
                const ref_modulus *sp = &key_mod[sp0];
                for (size_t l = 0; l < n; l++)
                    lazy[l] += (u128)negate_uint_mod(ref_barrett_reduce_63(ct[l], sp), sp->value);
            }
        }
        for (size_t l = 0; l < n; l++)
            temp[l] = ref_barrett_reduce_128((uint64_t)lazy[l], (uint64_t)(lazy[l] >> 64), qi);
        if (is_ckks)
            ref_ntt_forward_lazy(temp, &key_tables[i], strict);
        else
            ref_ntt_inverse_lazy(poly + i * n, &key_tables[i]);
        uint64_t P_qi = 1;
        for (size_t j = 0; j < n_sp_rns; j++)
            P_qi = ref_multiply_uint_mod(P_qi, key_mod[sp0 + j].value, qi);
        uint64_t invP;
        ref_try_invert_uint_mod(P_qi, qi->value, &invP);
        uint64_t invPs = ref_shoupify(invP, qi->value);
        uint64_t *row = poly + i * n;
        for (size_t l = 0; l < n; l++)
            row[l] = mulmod_shoup(row[l] + temp[l], invP, invPs, qi->value); /* PolyMulConstant::poly_add :63-74 */
    }
    free(lazy);
    free(temp);
}

/* ------------------------------------------------------------------------------------------
 * Context (context.cpp:455-540) -- plain-struct mirror: key level = all primes,
 * first ciphertext level = all - nsp, level addressed by k = number of leading primes.
 * ---------------------------------------------------------------------------------------- */
int ref_context_init(ref_context *c, int scheme, int logn, const uint64_t *key_moduli, size_t n_key, size_t nsp,
                     uint64_t t, int mode)
{
    memset(c, 0, sizeof(*c));
    if (nsp == 0 || n_key <= nsp)
        return -1;
    c->scheme = scheme;
    c->logn = logn;
    c->n = (size_t)1 << logn;
    c->n_key = n_key;
    c->nsp = nsp;
    c->k_first = n_key - nsp;
    c->t = t;
    c->mode = mode;
    c->key_mod = (ref_modulus *)calloc(n_key, sizeof(ref_modulus));
    c->key_tables = (ref_ntt_tables *)calloc(n_key, sizeof(ref_ntt_tables));
    c->rns_tools = (ref_rns_tool **)calloc(n_key + 1, sizeof(ref_rns_tool *));
    for (size_t i = 0; i < n_key; i++)
    {
        if (ref_modulus_init(&c->key_mod[i], key_moduli[i]))
            return -1;
        if (ref_ntt_tables_init(&c->key_tables[i], logn, key_moduli[i]))
            return -1;
    }
    return 0;
}

void ref_context_free(ref_context *c)
{
    if (c->key_tables)
        for (size_t i = 0; i < c->n_key; i++)
            ref_ntt_tables_free(&c->key_tables[i]);
    if (c->rns_tools)
        for (size_t i = 0; i <= c->n_key; i++)
            if (c->rns_tools[i])
            {
                ref_rns_tool_free(c->rns_tools[i]);
                free(c->rns_tools[i]);
            }
    free(c->key_mod);
    free(c->key_tables);
    free(c->rns_tools);
    memset(c, 0, sizeof(*c));
}

const ref_rns_tool *ref_context_rns_tool(ref_context *c, size_t k)
{
    if (k == 0 || k > c->n_key)
        return NULL;
    if (!c->rns_tools[k])
    {
        uint64_t q[64];
        for (size_t i = 0; i < k; i++)
            q[i] = c->key_mod[i].value;
        ref_rns_tool *rt = (ref_rns_tool *)malloc(sizeof(ref_rns_tool));
        if (ref_rns_tool_init(rt, c->n, q, k, c->scheme == REF_SCHEME_BFV ? c->t : 0))
        {
            free(rt);
            return NULL;
        }
        c->rns_tools[k] = rt;
    }
    return c->rns_tools[k];
}

/* ------------------------------------------------------------------------------------------
 * Evaluator drivers
 * ---------------------------------------------------------------------------------------- */
/* evaluator.cpp:274-445 */
int ref_bfv_multiply(ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                     uint64_t *out)
{
    const ref_rns_tool *rt = ref_context_rns_tool(c, k);
    if (!rt)
        return -1;
    const size_t n = c->n, B = rt->Bsk_size;
    const int strict = c->mode == REF_MODE_STRICT;
    size_t dest = sa + sb - 1;
    const uint64_t *in[2] = { a, b };
    size_t sz[2] = { sa, sb };
    uint64_t *xq[2], *xB[2];
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * (B + 1));
    for (int s = 0; s < 2; s++)
    {
        xq[s] = (uint64_t *)malloc(sizeof(uint64_t) * n * k * sz[s]);
        xB[s] = (uint64_t *)malloc(sizeof(uint64_t) * n * B * sz[s]);
        for (size_t pidx = 0; pidx < sz[s]; pidx++)
        {
            const uint64_t *src = in[s] + pidx * k * n;
            uint64_t *dq = xq[s] + pidx * k * n, *dB = xB[s] + pidx * B * n;
            /* :338-339 */
            memcpy(dq, src, sizeof(uint64_t) * n * k);
            for (size_t i = 0; i < k; i++)
                ref_ntt_forward_lazy(dq + i * n, &c->key_tables[i], strict);
            /* :345-352 */
            ref_fastbconv_m_tilde(rt, src, temp);
            ref_sm_mrq(rt, temp, dB);
            for (size_t i = 0; i < B; i++)
                ref_ntt_forward_lazy(dB + i * n, &rt->Bsk_ntt[i], strict);
        }
    }
    uint64_t *dq = (uint64_t *)calloc(n * k * dest, sizeof(uint64_t));
    uint64_t *dB = (uint64_t *)calloc(n * B * dest, sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * n);
    /* :376-420 */
    for (size_t I = 0; I < dest; I++)
    {
        size_t last1 = I < sa - 1 ? I : sa - 1;
        size_t first2 = I < sb - 1 ? I : sb - 1;
        size_t first1 = I - first2;
        size_t steps = last1 - first1 + 1;
        for (size_t s = 0; s < steps; s++)
        {
            size_t i1 = first1 + s, i2 = first2 - s;
            for (size_t r = 0; r < k; r++)
            {
                ref_dyadic_product_coeffmod(xq[0] + (i1 * k + r) * n, xq[1] + (i2 * k + r) * n, n, &c->key_mod[r],
                                            prod);
                ref_add_poly_coeffmod(prod, dq + (I * k + r) * n, n, &c->key_mod[r], dq + (I * k + r) * n);
            }
            for (size_t r = 0; r < B; r++)
            {
                ref_dyadic_product_coeffmod(xB[0] + (i1 * B + r) * n, xB[1] + (i2 * B + r) * n, n, &rt->Bsk[r], prod);
                ref_add_poly_coeffmod(prod, dB + (I * B + r) * n, n, &rt->Bsk[r], dB + (I * B + r) * n);
            }
        }
    }
    /* :423-424 */
    for (size_t I = 0; I < dest; I++)
    {
        for (size_t r = 0; r < k; r++)
            ref_ntt_inverse(dq + (I * k + r) * n, &c->key_tables[r]);
        for (size_t r = 0; r < B; r++)
            ref_ntt_inverse(dB + (I * B + r) * n, &rt->Bsk_ntt[r]);
    }
    /* :427-444 */
    uint64_t *tqB = (uint64_t *)malloc(sizeof(uint64_t) * n * (k + B));
    uint64_t *tB = (uint64_t *)malloc(sizeof(uint64_t) * n * B);
    for (size_t I = 0; I < dest; I++)
    {
        for (size_t r = 0; r < k; r++)
            ref_multiply_poly_scalar_coeffmod(dq + (I * k + r) * n, n, c->t, &c->key_mod[r], tqB + r * n);
        for (size_t r = 0; r < B; r++)
            ref_multiply_poly_scalar_coeffmod(dB + (I * B + r) * n, n, c->t, &rt->Bsk[r], tqB + (k + r) * n);
        ref_fast_floor(rt, tqB, tB);
        ref_fastbconv_sk(rt, tB, out + I * k * n);
    }
    free(tqB);
    free(tB);
    free(prod);
    free(dq);
    free(dB);
    free(temp);
    for (int s = 0; s < 2; s++)
    {
        free(xq[s]);
        free(xB[s]);
    }
    return 0;
}

/* evaluator.cpp:560-702 (bfv_square). A ciphertext of size != 2 takes the multiply path (:579-583); for size 2 the
 * two polynomials are lifted and transformed ONCE (:604-634), the tensor product is c_0 = a_0^2, c_1 = a_0 a_1 added to
 * itself, c_2 = a_1^2 (:644-657) over q and over Bsk, the inverse transforms are the canonicalising ones (:663-664), and the
 * floor / base-conversion tail (:667-701) is the one of bfv_multiply. */
int ref_bfv_square(ref_context *c, size_t k, const uint64_t *a, size_t sa, uint64_t *out)
{
    if (sa != 2)
        return ref_bfv_multiply(c, k, a, sa, a, sa, out); /* :579-583 */
    const ref_rns_tool *rt = ref_context_rns_tool(c, k);
    if (!rt)
        return -1;
    const size_t n = c->n, B = rt->Bsk_size, dest = 3;
    const int strict = c->mode == REF_MODE_STRICT;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * (B + 1));
    uint64_t *xq = (uint64_t *)malloc(sizeof(uint64_t) * n * k * 2);
    uint64_t *xB = (uint64_t *)malloc(sizeof(uint64_t) * n * B * 2);
    /* :604-634 behz_extend_base_convert_to_ntt over the two polynomials */
    for (size_t pidx = 0; pidx < 2; pidx++)
    {
        const uint64_t *src = a + pidx * k * n;
        uint64_t *dq = xq + pidx * k * n, *dB = xB + pidx * B * n;
        memcpy(dq, src, sizeof(uint64_t) * n * k);                      /* :607 set_poly */
        for (size_t i = 0; i < k; i++)
            ref_ntt_forward_lazy(dq + i * n, &c->key_tables[i], strict); /* :609 */
        ref_fastbconv_m_tilde(rt, src, temp);                           /* :615 */
        ref_sm_mrq(rt, temp, dB);                                       /* :618 */
        for (size_t i = 0; i < B; i++)
            ref_ntt_forward_lazy(dB + i * n, &rt->Bsk_ntt[i], strict);  /* :621 */
    }
    uint64_t *dq = (uint64_t *)calloc(n * k * dest, sizeof(uint64_t));
    uint64_t *dB = (uint64_t *)calloc(n * B * dest, sizeof(uint64_t));
    /* :644-660 behz_ciphertext_square over q, then over Bsk */
    for (size_t r = 0; r < k; r++)
    {
        const ref_modulus *m = &c->key_mod[r];
        const uint64_t *x0 = xq + r * n, *x1 = xq + (k + r) * n;
        ref_dyadic_product_coeffmod(x0, x0, n, m, dq + r * n);                                  /* :647 */
        ref_dyadic_product_coeffmod(x0, x1, n, m, dq + (k + r) * n);                            /* :650 */
        ref_add_poly_coeffmod(dq + (k + r) * n, dq + (k + r) * n, n, m, dq + (k + r) * n);      /* :651 */
        ref_dyadic_product_coeffmod(x1, x1, n, m, dq + (2 * k + r) * n);                        /* :654 */
    }
    for (size_t r = 0; r < B; r++)
    {
        const ref_modulus *m = &rt->Bsk[r];
        const uint64_t *x0 = xB + r * n, *x1 = xB + (B + r) * n;
        ref_dyadic_product_coeffmod(x0, x0, n, m, dB + r * n);
        ref_dyadic_product_coeffmod(x0, x1, n, m, dB + (B + r) * n);
        ref_add_poly_coeffmod(dB + (B + r) * n, dB + (B + r) * n, n, m, dB + (B + r) * n);
        ref_dyadic_product_coeffmod(x1, x1, n, m, dB + (2 * B + r) * n);
    }
    /* :663-664 (the canonicalising inverse, unlike bfv_multiply's :423-424 -- same residues) */
    for (size_t I = 0; I < dest; I++)
    {
        for (size_t r = 0; r < k; r++)
            ref_ntt_inverse(dq + (I * k + r) * n, &c->key_tables[r]);
        for (size_t r = 0; r < B; r++)
            ref_ntt_inverse(dB + (I * B + r) * n, &rt->Bsk_ntt[r]);
    }
    /* :667-701 */
    uint64_t *tqB = (uint64_t *)malloc(sizeof(uint64_t) * n * (k + B));
    uint64_t *tB = (uint64_t *)malloc(sizeof(uint64_t) * n * B);
    for (size_t I = 0; I < dest; I++)
    {
        for (size_t r = 0; r < k; r++)
            ref_multiply_poly_scalar_coeffmod(dq + (I * k + r) * n, n, c->t, &c->key_mod[r], tqB + r * n);
        for (size_t r = 0; r < B; r++)
            ref_multiply_poly_scalar_coeffmod(dB + (I * B + r) * n, n, c->t, &rt->Bsk[r], tqB + (k + r) * n);
        ref_fast_floor(rt, tqB, tB);
        ref_fastbconv_sk(rt, tB, out + I * k * n);
    }
    free(tqB);
    free(tB);
    free(dq);
    free(dB);
    free(temp);
    free(xq);
    free(xB);
    return 0;
}

/* evaluator.cpp:704-770 (ckks_square): size != 2 -> ckks_multiply (:720-724); else c_0 = a_0^2, c_1 = a_0 a_1 + a_0 a_1,
 * c_2 = a_1^2 (:752-760), copied back (:763). (The scale bookkeeping :727-733, 766 is host metadata.) */
int ref_ckks_square(ref_context *c, size_t k, const uint64_t *a, size_t sa, uint64_t *out)
{
    if (sa != 2)
        return ref_ckks_multiply(c, k, a, sa, a, sa, out);
    const size_t n = c->n;
    uint64_t *tmp = (uint64_t *)calloc(n * k * 3, sizeof(uint64_t));
    for (size_t r = 0; r < k; r++)
    {
        const ref_modulus *m = &c->key_mod[r];
        const uint64_t *x0 = a + r * n, *x1 = a + (k + r) * n;
        ref_dyadic_product_coeffmod(x0, x0, n, m, tmp + r * n);                                   /* :753 */
        ref_dyadic_product_coeffmod(x0, x1, n, m, tmp + (k + r) * n);                             /* :756 */
        ref_add_poly_coeffmod(tmp + (k + r) * n, tmp + (k + r) * n, n, m, tmp + (k + r) * n);     /* :757 */
        ref_dyadic_product_coeffmod(x1, x1, n, m, tmp + (2 * k + r) * n);                         /* :760 */
    }
    memcpy(out, tmp, sizeof(uint64_t) * n * k * 3); /* :763 */
    free(tmp);
    return 0;
}

/* evaluator.cpp:447-527 */
int ref_ckks_multiply(ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                      uint64_t *out)
{
    const size_t n = c->n;
    size_t dest = sa + sb - 1;
    uint64_t *tmp = (uint64_t *)calloc(n * k * dest, sizeof(uint64_t));
    uint64_t *prod = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t I = 0; I < dest; I++)
    {
        size_t last1 = I < sa - 1 ? I : sa - 1;
        size_t first2 = I < sb - 1 ? I : sb - 1;
        size_t first1 = I - first2;
        size_t steps = last1 - first1 + 1;
        for (size_t s = 0; s < steps; s++)
        {
            size_t i1 = first1 + s, i2 = first2 - s;
            for (size_t r = 0; r < k; r++)
            {
                ref_dyadic_product_coeffmod(a + (i1 * k + r) * n, b + (i2 * k + r) * n, n, &c->key_mod[r], prod);
                ref_add_poly_coeffmod(prod, tmp + (I * k + r) * n, n, &c->key_mod[r], tmp + (I * k + r) * n);
            }
        }
    }
    memcpy(out, tmp, sizeof(uint64_t) * n * k * dest);
    free(tmp);
    free(prod);
    return 0;
}

/* evaluator.cpp:2259-2368, with the split SURVEY 8(e) "latency mode" describes: the inner product over the digits
 * [j0, j1) only, its two 128-bit accumulators reduced to canonical residues and handed out (`partial_out`, 2 x rows x n) --
 * or, on the other side of the all-reduce, the sum of such partials taken in (`partial_in`, every word below
 * ranks * p < 2^63) and reduced once more before the reference's tail (:2351-2366) runs. Modular sums are associative: the
 * canonical residue of the summed partials is the canonical residue of the whole 128-bit sum, so every later word is the one
 * the unsplit function produces. Both NULL and [0, n_bundles): the function as the reference has it. */
static int switch_key_core(ref_context *c, size_t k, uint64_t *ct, const uint64_t *target, const uint64_t *key, size_t j0,
                           size_t j1, uint64_t *partial_out, const uint64_t *partial_in)
{
    const size_t n = c->n, n_ct = k, n_all = c->k_first, n_total = c->n_key, nsp = n_total - n_all;
    const size_t n_bundles = (n_ct + nsp - 1) / nsp;
    const int is_ckks = c->scheme == REF_SCHEME_CKKS;
    const int strict = c->mode == REF_MODE_STRICT;
    const size_t rows = n_ct + nsp;
    if (j1 > n_bundles || j0 > j1)
        return -1;
    u128 *lazy[2];
    lazy[0] = (u128 *)calloc(rows * n, sizeof(u128));
    lazy[1] = (u128 *)calloc(rows * n, sizeof(u128));
    uint64_t *ext = (uint64_t *)malloc(sizeof(uint64_t) * rows * n);
    for (size_t j = j0; j < j1 && !partial_in; j++)
    {
        size_t rns0 = j * nsp;
        size_t rns1 = rns0 + nsp < n_ct ? rns0 + nsp : n_ct;
        /* :2302-2307 */
        for (size_t r = rns0; r < rns1; r++)
        {
            memcpy(ext + r * n, target + r * n, sizeof(uint64_t) * n);
            if (is_ckks)
                ref_ntt_inverse(ext + r * n, &c->key_tables[r]);
        }
        /* :2310 */
        ref_modup_rns(ext + rns0 * n, ext, n, n_ct, nsp, j, c->key_mod, n_total);
        /* :2315-2335 */
        for (size_t r = 0; r < rows; r++)
        {
            int is_sp = r >= n_ct;
            size_t rns_idx = is_sp ? n_all + r - n_ct : r;
            const uint64_t *ctp;
            uint64_t *scratch = NULL;
            if (r >= rns0 && r < rns1)
            {
                if (strict && !is_ckks)
                {
                    /* STRICT: NTT the coefficient-form in-bundle row before the inner product */
                    scratch = (uint64_t *)malloc(sizeof(uint64_t) * n);
                    memcpy(scratch, ext + r * n, sizeof(uint64_t) * n);
                    ref_ntt_forward_lazy(scratch, &c->key_tables[rns_idx], 1);
                    ctp = scratch;
                }
                else
                    ctp = target + r * n; /* F3: used as-is */
            }
            else
            {
                ref_ntt_forward_lazy(ext + r * n, &c->key_tables[rns_idx], strict);
                ctp = ext + r * n;
            }
            for (int l = 0; l < 2; l++)
            {
                const uint64_t *kp = key + ((j * 2 + (size_t)l) * n_total + rns_idx) * n;
                u128 *acc = lazy[l] + r * n;
                for (size_t d = 0; d < n; d++)
                    acc[d] += (u128)ctp[d] * kp[d];
            }
            free(scratch);
        }
    }
    for (int b = 0; b < 2; b++)
    {
        /* :2341-2357 */
        for (size_t r = 0; r < rows; r++)
        {
            int is_sp = r >= n_ct;
            size_t rns_idx = is_sp ? n_all + r - n_ct : r;
            const u128 *acc = lazy[b] + r * n;
            uint64_t *dst = partial_out ? partial_out + ((size_t)b * rows + r) * n : ext + r * n;
            if (partial_in) /* latency mode, after the all-reduce: the summed canonical partials, reduced */
                for (size_t l = 0; l < n; l++)
                    dst[l] = ref_barrett_reduce_63(partial_in[((size_t)b * rows + r) * n + l], &c->key_mod[rns_idx]);
            else
                for (size_t l = 0; l < n; l++)
                    dst[l] = ref_barrett_reduce_128((uint64_t)acc[l], (uint64_t)(acc[l] >> 64), &c->key_mod[rns_idx]);
            if (partial_out)
                continue; /* latency mode, before the all-reduce: the reduced partial leaves here */
            if (is_sp)
                ref_ntt_inverse_lazy(dst, &c->key_tables[rns_idx]);
        }
        if (partial_out)
            continue;
        /* :2361 */
        ref_rescale_special_rns_inplace(ext, is_ckks, n, n_ct, nsp, c->key_mod, n_total, c->key_tables, strict);
        /* :2363-2366 */
        uint64_t *enc = ct + (size_t)b * n_ct * n;
        for (size_t i = 0; i < n_ct; i++)
            ref_add_poly_coeffmod(ext + i * n, enc + i * n, n, &c->key_mod[i], enc + i * n);
    }
    free(lazy[0]);
    free(lazy[1]);
    free(ext);
    return 0;
}

int ref_switch_key_inplace(ref_context *c, size_t k, uint64_t *ct, const uint64_t *target, const uint64_t *key)
{
    const size_t nsp = c->n_key - c->k_first;
    return switch_key_core(c, k, ct, target, key, 0, (k + nsp - 1) / nsp, NULL, NULL);
}

/* SURVEY 8(e) latency mode: the digits [j0, j1) of one key switch -> 2 x (k + nsp) x n canonical partial products */
int ref_switch_key_partial(ref_context *c, size_t k, const uint64_t *target, const uint64_t *key, size_t j0, size_t j1,
                           uint64_t *partial)
{
    return switch_key_core(c, k, NULL, target, key, j0, j1, partial, NULL);
}

/* ... and the rest of the key switch on the element-wise SUM of every rank's partials (words below 2^63) */
int ref_switch_key_finish(ref_context *c, size_t k, uint64_t *ct, const uint64_t *partial_sum)
{
    return switch_key_core(c, k, ct, NULL, NULL, 0, 0, NULL, partial_sum);
}

/* evaluator.cpp:772-827: the target is always the LAST polynomial (the iterator is never moved). */
int ref_relinearize(ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *const *keys)
{
    if (size < 2)
        return -1;
    for (size_t I = 0; I + 2 < size; I++)
    {
        size_t key_power = size - 1 - I;
        int rc = ref_switch_key_inplace(c, k, ct, ct + (size - 1) * k * c->n, keys[key_power - 2]);
        if (rc)
            return rc;
    }
    return 0;
}

/* evaluator.cpp:829-892 */
int ref_mod_switch_scale_to_next(ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out)
{
    if (k < 2)
        return -1;
    const ref_rns_tool *rt = ref_context_rns_tool(c, k);
    if (!rt)
        return -1;
    const size_t n = c->n;
    uint64_t *copy = (uint64_t *)malloc(sizeof(uint64_t) * n * k);
    for (size_t s = 0; s < size; s++)
    {
        memcpy(copy, ct + s * k * n, sizeof(uint64_t) * n * k);
        if (c->scheme == REF_SCHEME_BFV)
            ref_divide_and_round_q_last_inplace(rt, copy);
        else
            ref_divide_and_round_q_last_ntt_inplace(rt, copy, c->key_tables, c->mode == REF_MODE_STRICT);
        memcpy(out + s * (k - 1) * n, copy, sizeof(uint64_t) * n * (k - 1));
    }
    free(copy);
    return 0;
}

/* evaluator.cpp:894-957 */
int ref_mod_switch_drop_to_next(ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out)
{
    if (k < 2)
        return -1;
    const size_t n = c->n;
    for (size_t s = 0; s < size; s++)
        memmove(out + s * (k - 1) * n, ct + s * k * n, sizeof(uint64_t) * n * (k - 1));
    return 0;
}

/* evaluator.cpp:1841-1943 */
int ref_apply_galois_inplace(ref_context *c, size_t k, uint64_t *ct, uint32_t galois_elt, const uint64_t *key)
{
    const size_t n = c->n;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * k);
    for (int s = 0; s < 2; s++)
    {
        uint64_t *poly = ct + (size_t)s * k * n;
        for (size_t r = 0; r < k; r++)
        {
            if (c->scheme == REF_SCHEME_BFV)
                ref_apply_galois(poly + r * n, c->logn, galois_elt, &c->key_mod[r], temp + r * n);
            else
                ref_apply_galois_ntt(poly + r * n, c->logn, galois_elt, temp + r * n);
        }
        if (s == 0)
            memcpy(poly, temp, sizeof(uint64_t) * n * k);
    }
    memset(ct + k * n, 0, sizeof(uint64_t) * n * k);
    int rc = ref_switch_key_inplace(c, k, ct, temp, key);
    free(temp);
    return rc;
}

/* ------------------------------------------------------------------------------------------
 * Evaluator negate / add / sub / multiply_plain, is_transparent
 * ---------------------------------------------------------------------------------------- */
void ref_evaluator_negate(const ref_context *c, size_t k, const uint64_t *ct, size_t size, uint64_t *out)
{
    const size_t n = c->n;
    for (size_t j = 0; j < size; j++)
        for (size_t r = 0; r < k; r++) /* evaluator.cpp:80 */
            ref_negate_poly_coeffmod(ct + (j * k + r) * n, n, &c->key_mod[r], out + (j * k + r) * n);
}

static void add_sub(const ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                    uint64_t *out, int sub)
{
    const size_t n = c->n, pw = k * n;
    const size_t mn = sa < sb ? sa : sb, mx = sa < sb ? sb : sa;
    for (size_t j = 0; j < mn; j++)
        for (size_t r = 0; r < k; r++) /* evaluator.cpp:135 / :213 */
        {
            if (sub)
                ref_sub_poly_coeffmod(a + j * pw + r * n, b + j * pw + r * n, n, &c->key_mod[r], out + j * pw + r * n);
            else
                ref_add_poly_coeffmod(a + j * pw + r * n, b + j * pw + r * n, n, &c->key_mod[r], out + j * pw + r * n);
        }
    for (size_t j = mn; j < mx; j++)
    {
        if (sa > sb) /* encrypted1 keeps its own tail */
            memmove(out + j * pw, a + j * pw, sizeof(uint64_t) * pw);
        else if (!sub) /* :138-143 */
            memmove(out + j * pw, b + j * pw, sizeof(uint64_t) * pw);
        else /* :216-220 */
            for (size_t r = 0; r < k; r++)
                ref_negate_poly_coeffmod(b + j * pw + r * n, n, &c->key_mod[r], out + j * pw + r * n);
    }
}

void ref_evaluator_add(const ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                       uint64_t *out)
{
    add_sub(c, k, a, sa, b, sb, out, 0);
}

void ref_evaluator_sub(const ref_context *c, size_t k, const uint64_t *a, size_t sa, const uint64_t *b, size_t sb,
                       uint64_t *out)
{
    add_sub(c, k, a, sa, b, sb, out, 1);
}

void ref_multiply_plain_ntt(const ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *plain_ntt)
{
    const size_t n = c->n;
    for (size_t j = 0; j < size; j++) /* evaluator.cpp:1638-1641 */
        for (size_t r = 0; r < k; r++)
            ref_dyadic_product_coeffmod(ct + (j * k + r) * n, plain_ntt + r * n, n, &c->key_mod[r], ct + (j * k + r) * n);
}

int ref_multiply_plain(const ref_context *c, size_t k, uint64_t *ct, size_t size, const uint64_t *plain)
{
    const size_t n = c->n;
    for (size_t r = 0; r < k; r++)
        if (c->key_mod[r].value <= c->t) /* context.cpp:297-301: using_fast_plain_lift */
            return -1;
    const uint64_t threshold = (c->t + 1) >> 1; /* context.cpp:322 */
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n * k);
    for (size_t r = 0; r < k; r++)
    {
        const uint64_t inc = c->key_mod[r].value - c->t; /* context.cpp:331 */
        for (size_t i = 0; i < n; i++) /* evaluator.cpp:1583-1592 */
            temp[r * n + i] = plain[i] + (inc & (uint64_t)(-(int64_t)(plain[i] >= threshold)));
        ref_ntt_forward(temp + r * n, &c->key_tables[r], 0); /* :1596-1597 */
    }
    for (size_t j = 0; j < size; j++) /* :1599-1606 */
        for (size_t r = 0; r < k; r++)
        {
            uint64_t *row = ct + (j * k + r) * n;
            ref_ntt_forward_lazy(row, &c->key_tables[r], c->mode == REF_MODE_STRICT);
            ref_dyadic_product_coeffmod(row, temp + r * n, n, &c->key_mod[r], row);
            ref_ntt_inverse(row, &c->key_tables[r]);
        }
    free(temp);
    return 0;
}

int ref_is_transparent(const ref_context *c, size_t k, const uint64_t *ct, size_t size)
{
    if (size < 2)
        return 1;
    const size_t pw = k * c->n;
    for (size_t i = pw; i < size * pw; i++)
        if (ct[i])
            return 0;
    return 1;
}

/* ------------------------------------------------------------------------------------------
 * encrypt / key generation / decrypt restated for the semantic end-to-end check
 * ---------------------------------------------------------------------------------------- */
void ref_sample_ternary(int8_t *s, size_t n, uint64_t *state)
{
    for (size_t i = 0; i < n; i++)
        s[i] = (int8_t)(ref_splitmix64(state) % 3) - 1;
}

void ref_sample_noise(int8_t *e, size_t n, uint64_t *state)
{
    for (size_t i = 0; i < n; i++)
    {
        const uint64_t r = ref_splitmix64(state);
        e[i] = (int8_t)(__builtin_popcountll(r & 0x1FFFFFull) - __builtin_popcountll((r >> 21) & 0x1FFFFFull)); /* var 10.5 */
    }
}

void ref_small_poly_to_rns(const ref_context *c, const int8_t *s, size_t rows, int to_ntt, uint64_t *out)
{
    const size_t n = c->n;
    for (size_t r = 0; r < rows; r++)
    {
        const uint64_t p = c->key_mod[r].value;
        for (size_t i = 0; i < n; i++)
            out[r * n + i] = s[i] >= 0 ? (uint64_t)s[i] : p - (uint64_t)(-s[i]);
        if (to_ntt)
            ref_ntt_forward(out + r * n, &c->key_tables[r], 0);
    }
}

static void small32_to_rns(const ref_context *c, const int32_t *s, size_t rows, uint64_t *out)
{
    const size_t n = c->n;
    for (size_t r = 0; r < rows; r++) /* the representation sample_poly_ternary/normal write, util/rlwe.cpp:25-95 */
    {
        const uint64_t p = c->key_mod[r].value;
        for (size_t i = 0; i < n; i++)
            out[r * n + i] = s[i] >= 0 ? (uint64_t)s[i] : p - (uint64_t)(-(int64_t)s[i]);
    }
}

void ref_encrypt_zero_symmetric_given(const ref_context *c, size_t rows, const uint64_t *sk_ntt, int is_ntt_form,
                                      const uint64_t *a_ntt, const int32_t *e, uint64_t *ct)
{
    const size_t n = c->n;
    uint64_t *c0 = ct, *c1 = ct + rows * n;
    uint64_t *noise = (uint64_t *)malloc(sizeof(uint64_t) * rows * n);
    if (c1 != a_ntt)
        memcpy(c1, a_ntt, sizeof(uint64_t) * rows * n); /* rlwe.cpp:245-249: a sampled directly in NTT form */
    small32_to_rns(c, e, rows, noise);
    for (size_t r = 0; r < rows; r++) /* rlwe.cpp:266-284 */
    {
        const ref_modulus *m = &c->key_mod[r];
        ref_dyadic_product_coeffmod(sk_ntt + r * n, c1 + r * n, n, m, c0 + r * n);
        if (is_ntt_form)
            ref_ntt_forward(noise + r * n, &c->key_tables[r], 0);
        else
            ref_ntt_inverse(c0 + r * n, &c->key_tables[r]);
        ref_add_poly_coeffmod(noise + r * n, c0 + r * n, n, m, c0 + r * n);
        ref_negate_poly_coeffmod(c0 + r * n, n, m, c0 + r * n);
        if (!is_ntt_form) /* :286-293 */
            ref_ntt_inverse(c1 + r * n, &c->key_tables[r]);
    }
    free(noise);
}

void ref_encrypt_zero_symmetric(const ref_context *c, size_t rows, const uint64_t *sk_ntt, int is_ntt_form,
                                uint64_t *state, uint64_t *ct)
{
    const size_t n = c->n;
    uint64_t *c1 = ct + rows * n;
    int8_t *e = (int8_t *)malloc(n);
    int32_t *e32 = (int32_t *)malloc(sizeof(int32_t) * n);
    for (size_t r = 0; r < rows; r++)
        for (size_t i = 0; i < n; i++)
            c1[r * n + i] = ref_splitmix64(state) % c->key_mod[r].value;
    ref_sample_noise(e, n, state);
    for (size_t i = 0; i < n; i++)
        e32[i] = e[i];
    ref_encrypt_zero_symmetric_given(c, rows, sk_ntt, is_ntt_form, c1, e32, ct);
    free(e);
    free(e32);
}

/* util/rlwe.cpp:140-202 with the samples handed in: u ternary, e = two noise polynomials (size-2 public key);
 * pk = 2 x rows x N in NTT form; ct[j] = pk[j] * u + e[j] */
void ref_encrypt_zero_asymmetric_given(const ref_context *c, size_t rows, const uint64_t *pk, int is_ntt_form,
                                       const int32_t *u, const int32_t *e, uint64_t *ct)
{
    const size_t n = c->n;
    uint64_t *tmp = (uint64_t *)malloc(sizeof(uint64_t) * rows * n);
    small32_to_rns(c, u, rows, tmp);
    for (size_t r = 0; r < rows; r++) /* :168-185 */
    {
        const ref_modulus *m = &c->key_mod[r];
        ref_ntt_forward(tmp + r * n, &c->key_tables[r], 0);
        for (size_t j = 0; j < 2; j++)
        {
            uint64_t *dst = ct + (j * rows + r) * n;
            ref_dyadic_product_coeffmod(tmp + r * n, pk + (j * rows + r) * n, n, m, dst);
            if (!is_ntt_form)
                ref_ntt_inverse(dst, &c->key_tables[r]);
        }
    }
    for (size_t j = 0; j < 2; j++) /* :187-201 */
    {
        small32_to_rns(c, e + j * n, rows, tmp);
        for (size_t r = 0; r < rows; r++)
        {
            uint64_t *dst = ct + (j * rows + r) * n;
            if (is_ntt_form)
                ref_ntt_forward(tmp + r * n, &c->key_tables[r], 0);
            ref_add_poly_coeffmod(tmp + r * n, dst, n, &c->key_mod[r], dst);
        }
    }
    free(tmp);
}

/* q = prod of the first k key primes as little-endian 64-bit limbs */
static void big_product(const ref_context *c, size_t k, uint64_t *limbs /* [k] */)
{
    memset(limbs, 0, sizeof(uint64_t) * k);
    limbs[0] = 1;
    for (size_t i = 0; i < k; i++)
    {
        unsigned __int128 carry = 0;
        for (size_t l = 0; l < k; l++)
        {
            const unsigned __int128 v = (unsigned __int128)limbs[l] * c->key_mod[i].value + carry;
            limbs[l] = (uint64_t)v;
            carry = v >> 64;
        }
    }
}
/* quotient (k limbs) and remainder of a k-limb number by a word */
static uint64_t big_divide_word(const uint64_t *num, size_t k, uint64_t d, uint64_t *quot)
{
    unsigned __int128 rem = 0;
    for (size_t l = k; l-- > 0;)
    {
        const unsigned __int128 cur = (rem << 64) | num[l];
        quot[l] = (uint64_t)(cur / d);
        rem = cur % d;
    }
    return (uint64_t)rem;
}
static uint64_t big_mod_word(const uint64_t *num, size_t k, uint64_t d)
{
    unsigned __int128 rem = 0;
    for (size_t l = k; l-- > 0;)
        rem = ((rem << 64) | num[l]) % d;
    return (uint64_t)rem;
}

/* util/scalingvariant.cpp:15-52 (sub = 0) and :54-92 (sub != 0): c0 +-= round(q * plain / t) over the first k primes;
 * plain = n coefficients < t, c0 = k x N */
void ref_multiply_add_plain_with_scaling_variant(const ref_context *c, size_t k, const uint64_t *plain, int sub,
                                                 uint64_t *c0)
{
    const size_t n = c->n;
    const uint64_t t = c->t;
    /* context.cpp:303-321: coeff_div_plain_modulus = floor(q / t) in RNS form, q mod t, (t + 1) / 2 */
    uint64_t q[64], quot[64];
    big_product(c, k, q);
    const uint64_t q_mod_t = big_divide_word(q, k, t, quot);
    const uint64_t threshold = (t + 1) >> 1;
    for (size_t i = 0; i < n; i++) /* scalingvariant.cpp:31-51 */
    {
        const unsigned __int128 numerator = (unsigned __int128)plain[i] * q_mod_t + threshold;
        const uint64_t fix = (uint64_t)(numerator / t);
        for (size_t j = 0; j < k; j++)
        {
            const ref_modulus *m = &c->key_mod[j];
            const uint64_t div_j = big_mod_word(quot, k, m->value);
            const unsigned __int128 z = (unsigned __int128)div_j * plain[i] + fix; /* multiply_add_uint_mod */
            const uint64_t scaled = ref_barrett_reduce_128((uint64_t)z, (uint64_t)(z >> 64), m);
            if (sub)
            {
                const uint64_t x = c0[j * n + i]; /* sub_uint64_mod */
                c0[j * n + i] = x >= scaled ? x - scaled : x + m->value - scaled;
            }
            else
            {
                const uint64_t sum = scaled + c0[j * n + i];
                c0[j * n + i] = sum >= m->value ? sum - m->value : sum;
            }
        }
    }
}

void ref_bfv_encrypt_symmetric(const ref_context *c, size_t k, const uint64_t *sk_ntt, const uint64_t *plain,
                               uint64_t *state, uint64_t *ct)
{
    ref_encrypt_zero_symmetric(c, k, sk_ntt, 0, state, ct);
    ref_multiply_add_plain_with_scaling_variant(c, k, plain, 0, ct);
}

/* ------------------------------------------------------------------------------------------
 * BatchEncoder (batchencoder.cpp). plain_tables = NTTTables(logn, t) (context.cpp:262-275).
 * ---------------------------------------------------------------------------------------- */
static uint32_t bitrev32(uint32_t x, int bits)
{
    uint32_t r = 0;
    for (int i = 0; i < bits; i++)
        r |= ((x >> i) & 1u) << (bits - 1 - i);
    return r;
}

/* batchencoder.cpp:70-94 */
void ref_batch_index_map(int logn, uint32_t *map)
{
    const size_t n = (size_t)1 << logn, row = n >> 1, m = n << 1;
    uint64_t pos = 1;
    for (size_t i = 0; i < row; i++)
    {
        map[i] = bitrev32((uint32_t)((pos - 1) >> 1), logn);
        map[row | i] = bitrev32((uint32_t)((m - pos - 1) >> 1), logn);
        pos = (pos * 3) & (m - 1);
    }
}

/* batchencoder.cpp:113-154: values (count <= n, each < t) -> plaintext coefficients */
void ref_batch_encode(const ref_ntt_tables *plain_tables, const uint64_t *values, size_t count, uint64_t *plain)
{
    const size_t n = plain_tables->n;
    uint32_t *map = (uint32_t *)malloc(sizeof(uint32_t) * n);
    ref_batch_index_map(plain_tables->logn, map);
    for (size_t i = 0; i < n; i++)
        plain[map[i]] = i < count ? values[i] : 0;
    ref_ntt_inverse(plain, plain_tables);
    free(map);
}

/* batchencoder.cpp:156-198: the int64 overload -- a negative value v is stored as t + v */
void ref_batch_encode_signed(const ref_ntt_tables *plain_tables, const int64_t *values, size_t count, uint64_t *plain)
{
    const size_t n = plain_tables->n;
    const uint64_t t = plain_tables->mod.value;
    uint64_t *u = (uint64_t *)calloc(n, sizeof(uint64_t));
    for (size_t i = 0; i < count && i < n; i++)
        u[i] = values[i] < 0 ? t + (uint64_t)values[i] : (uint64_t)values[i];
    ref_batch_encode(plain_tables, u, count, plain);
    free(u);
}

/* batchencoder.cpp:378-420: the int64 overload -- slot values above t/2 come back negative */
void ref_batch_decode_signed(const ref_ntt_tables *plain_tables, const uint64_t *plain, size_t count, int64_t *values)
{
    const size_t n = plain_tables->n;
    const uint64_t t = plain_tables->mod.value, half = t >> 1;
    uint64_t *u = (uint64_t *)calloc(n, sizeof(uint64_t));
    ref_batch_decode(plain_tables, plain, count, u);
    for (size_t i = 0; i < n; i++)
        values[i] = u[i] > half ? (int64_t)u[i] - (int64_t)t : (int64_t)u[i];
    free(u);
}

/* batchencoder.cpp:339-376: plaintext coefficients (count <= n) -> n values */
void ref_batch_decode(const ref_ntt_tables *plain_tables, const uint64_t *plain, size_t count, uint64_t *values)
{
    const size_t n = plain_tables->n;
    uint32_t *map = (uint32_t *)malloc(sizeof(uint32_t) * n);
    uint64_t *tmp = (uint64_t *)calloc(n, sizeof(uint64_t));
    ref_batch_index_map(plain_tables->logn, map);
    memcpy(tmp, plain, sizeof(uint64_t) * (count < n ? count : n));
    ref_ntt_forward(tmp, plain_tables, 0);
    for (size_t i = 0; i < n; i++)
        values[i] = tmp[map[i]];
    free(tmp);
    free(map);
}

/* ------------------------------------------------------------------------------------------
 * CKKSEncoder (ckks.cpp:14-77, ckks.h:405-747). Double-precision FFT; every floating-point operation
 * is written in the order the reference's std::complex<double> arithmetic performs it (no contraction, no re-association),
 * so a second implementation that keeps this order produces the same bits.
 * ---------------------------------------------------------------------------------------- */
static void croot(size_t m, size_t index, double *re, double *im) /* util/croots.cpp:42-70 */
{
    static const double PI_ = 3.1415926535897932384626433832795028842;
    index &= m - 1;
    if (index <= m / 8)
    {
        const double th = 2 * PI_ * (double)index / (double)m; /* std::polar(1.0, th), croots.cpp:35-39 */
        sincos(th, im, re); /* what gcc -O3 makes of the cos/sin pair; differs from sin()/cos() in a few last bits */
    }
    else if (index <= m / 4)
    {
        double a, b;
        croot(m, m / 4 - index, &a, &b);
        *re = b;
        *im = a;
    }
    else if (index <= m / 2)
    {
        double a, b;
        croot(m, m / 2 - index, &a, &b);
        *re = -a;
        *im = b; /* -conj */
    }
    else if (index <= 3 * m / 4)
    {
        double a, b;
        croot(m, index - m / 2, &a, &b);
        *re = -a;
        *im = -b;
    }
    else
    {
        double a, b;
        croot(m, m - index, &a, &b);
        *re = a;
        *im = -b;
    }
}

int ref_ckks_encoder_init(ref_ckks_encoder *enc, int logn)
{
    const size_t n = (size_t)1 << logn, slots = n >> 1, m = n << 1;
    if (logn < 2)
        return -1;
    enc->logn = logn;
    enc->n = n;
    enc->index_map = (uint32_t *)malloc(sizeof(uint32_t) * n);
    enc->roots = (double *)malloc(sizeof(double) * 2 * n);
    enc->inv_roots = (double *)malloc(sizeof(double) * 2 * n);
    uint64_t pos = 1;
    for (size_t i = 0; i < slots; i++) /* ckks.cpp:39-56, generator 5 */
    {
        enc->index_map[i] = bitrev32((uint32_t)((pos - 1) >> 1), logn);
        enc->index_map[slots | i] = bitrev32((uint32_t)((m - pos - 1) >> 1), logn);
        pos = (pos * 5) & (m - 1);
    }
    for (size_t i = 0; i < n; i++) /* :62-69 */
    {
        double re, im;
        croot(m, bitrev32((uint32_t)i, logn), &re, &im);
        enc->roots[2 * i] = re;
        enc->roots[2 * i + 1] = im;
        enc->inv_roots[2 * i] = re;
        enc->inv_roots[2 * i + 1] = -im;
    }
    return 0;
}

void ref_ckks_encoder_free(ref_ckks_encoder *enc)
{
    free(enc->index_map);
    free(enc->roots);
    free(enc->inv_roots);
}

/* significant bits of q_0 * ... * q_{rows-1} (context.cpp:178) */
static int total_bit_count(const ref_context *c, size_t rows)
{
    uint64_t q[64];
    big_product(c, rows, q);
    for (size_t l = rows; l-- > 0;)
        if (q[l])
            return (int)(64 * l) + 64 - __builtin_clzll(q[l]);
    return 0;
}

/* CKKSEncoder::encode_internal(double value, ...), ckks.cpp:80-216, branch by branch. out: rows x n (every coefficient of
 * row j holds the residue). returns 0, -1 scale out of bounds, -2 encoded value is too large */
int ref_ckks_encode_value(const ref_context *c, size_t rows, double value, double scale, uint64_t *out)
{
    const size_t n = c->n;
    const int total_bits = total_bit_count(c, rows);
    if (scale <= 0 || ((int)log2(scale) >= total_bits)) /* :105-109 */
        return -1;
    value *= scale;
    const int coeff_bit_count = (int)log2(fabs(value)) + 2;
    if (coeff_bit_count >= total_bits) /* :114-118 */
        return -2;
    const double two_pow_64 = pow(2.0, 64);
    double coeffd = round(value);
    const int is_negative = signbit(coeffd);
    coeffd = fabs(coeffd);
    for (size_t j = 0; j < rows; j++)
    {
        const uint64_t q = c->key_mod[j].value;
        uint64_t r;
        if (coeff_bit_count <= 64) /* :131-151 */
            r = (uint64_t)fabs(coeffd) % q;
        else if (coeff_bit_count <= 128) /* :152-176 */
        {
            const unsigned __int128 v =
                ((unsigned __int128)(uint64_t)(coeffd / two_pow_64) << 64) | (uint64_t)fmod(coeffd, two_pow_64);
            r = (uint64_t)(v % q);
        }
        else /* :177-209: limbs by repeated fmod / division, then the multi-precision remainder */
        {
            uint64_t limbs[64] = { 0 };
            size_t nl = 0;
            double d = coeffd;
            while (d >= 1 && nl < 64)
            {
                limbs[nl++] = (uint64_t)fmod(d, two_pow_64);
                d /= two_pow_64;
            }
            unsigned __int128 acc = 0;
            for (size_t l = nl; l-- > 0;)
                acc = ((acc << 64) | limbs[l]) % q;
            r = (uint64_t)acc;
        }
        if (is_negative)
            r = r ? q - r : 0; /* negate_uint_mod */
        for (size_t i = 0; i < n; i++)
            out[j * n + i] = r;
    }
    return 0;
}

/* CKKSEncoder::encode_internal(int64_t value, ...), ckks.cpp:218-275. returns 0, -2 encoded value is too large */
int ref_ckks_encode_int64(const ref_context *c, size_t rows, int64_t value, uint64_t *out)
{
    const size_t n = c->n;
    const uint64_t mag = value < 0 ? (uint64_t)(-(value + 1)) + 1 : (uint64_t)value;
    const int bits = mag ? 64 - __builtin_clzll(mag) : 0;
    if (bits + 2 >= total_bit_count(c, rows))
        return -2;
    for (size_t j = 0; j < rows; j++)
    {
        const uint64_t q = c->key_mod[j].value;
        uint64_t tmp = (uint64_t)value;
        if (value < 0)
            tmp += q; /* :254-257, wrapping as written */
        tmp %= q;
        for (size_t i = 0; i < n; i++)
            out[j * n + i] = tmp;
    }
    return 0;
}

/* ckks.h:405-617. values: n_values complex numbers (re, im interleaved), n_values <= n/2; out: rows x n, NTT form.
 * returns 0, -1 scale out of bounds, -2 encoded values are too large */
int ref_ckks_encode(const ref_context *c, const ref_ckks_encoder *enc, size_t rows, const double *values, size_t n_values,
                    double scale, uint64_t *out)
{
    const size_t n = enc->n, slots = n >> 1;
    const int logn = enc->logn, total_bits = total_bit_count(c, rows);
    if (scale <= 0 || ((int)log2(scale) + 1 >= total_bits))
        return -1;
    double *cv = (double *)calloc(2 * n, sizeof(double));
    for (size_t i = 0; i < n_values; i++) /* :452-456 */
    {
        cv[2 * enc->index_map[i]] = values[2 * i];
        cv[2 * enc->index_map[i] + 1] = values[2 * i + 1];
        cv[2 * enc->index_map[i + slots]] = values[2 * i];
        cv[2 * enc->index_map[i + slots] + 1] = -values[2 * i + 1];
    }
    size_t tt = 1;
    for (int i = 0; i < logn; i++) /* :458-482 */
    {
        const size_t mm = (size_t)1 << (logn - i), h = mm / 2;
        size_t k_start = 0;
        for (size_t j = 0; j < h; j++)
        {
            const double sr = enc->inv_roots[2 * (h + j)], si = enc->inv_roots[2 * (h + j) + 1];
            for (size_t k = k_start; k < k_start + tt; k++)
            {
                const double ur = cv[2 * k], ui = cv[2 * k + 1], vr = cv[2 * (k + tt)], vi = cv[2 * (k + tt) + 1];
                const double dr = ur - vr, di = ui - vi;
                cv[2 * k] = ur + vr;
                cv[2 * k + 1] = ui + vi;
                cv[2 * (k + tt)] = dr * sr - di * si;
                cv[2 * (k + tt) + 1] = dr * si + di * sr;
            }
            k_start += 2 * tt;
        }
        tt *= 2;
    }
    double n_inv = 1.0 / (double)n;
    n_inv *= scale;
    int max_bits = 1;
    for (size_t i = 0; i < n; i++) /* :489-500 */
    {
        cv[2 * i] *= n_inv;
        cv[2 * i + 1] *= n_inv;
        const double d = fmax(fabs(cv[2 * i]), 1.0);
        const int b = (int)log2(d) + 2;
        if (b > max_bits)
            max_bits = b;
    }
    if (max_bits >= total_bits)
    {
        free(cv);
        return -2;
    }
    const double two_pow_64 = 18446744073709551616.0;
    for (size_t i = 0; i < n; i++)
    {
        double coeffd = round(cv[2 * i]);
        const int negative = signbit(coeffd) != 0;
        coeffd = fabs(coeffd);
        for (size_t j = 0; j < rows; j++)
        {
            const ref_modulus *m = &c->key_mod[j];
            uint64_t r;
            if (max_bits <= 64) /* :515-540 */
                r = (uint64_t)coeffd % m->value;
            else if (max_bits <= 128) /* :541-568 */
                r = ref_barrett_reduce_128((uint64_t)fmod(coeffd, two_pow_64), (uint64_t)(coeffd / two_pow_64), m);
            else /* :569-607 with RNSBase::decompose (rns.cpp:292-325) */
            {
                uint64_t limbs[64] = { 0 };
                size_t nl = 0;
                for (double x = coeffd; x >= 1; x /= two_pow_64)
                    limbs[nl++] = (uint64_t)fmod(x, two_pow_64);
                uint64_t hi = limbs[rows - 1];
                if (rows > 1)
                    for (size_t l = rows - 1; l--;)
                        hi = ref_barrett_reduce_128(limbs[l], hi, m);
                else
                    hi %= m->value;
                r = hi;
            }
            out[j * n + i] = negative ? (r ? m->value - r : 0) : r; /* negate_uint_mod */
        }
    }
    for (size_t j = 0; j < rows; j++) /* :609-613 */
        ref_ntt_forward(out + j * n, &c->key_tables[j], 0);
    free(cv);
    return 0;
}

/* ckks.h:623-747. plain: rows x n in NTT form; values: n/2 complex numbers out. returns 0 / -1 scale out of bounds */
int ref_ckks_decode(const ref_context *c, const ref_ckks_encoder *enc, size_t rows, const uint64_t *plain, double scale,
                    double *values)
{
    const size_t n = enc->n, slots = n >> 1;
    const int logn = enc->logn;
    if (scale <= 0 || ((int)log2(scale) >= total_bit_count(c, rows)))
        return -1;
    const double inv_scale = 1.0 / scale;
    uint64_t *copy = (uint64_t *)malloc(sizeof(uint64_t) * rows * n);
    memcpy(copy, plain, sizeof(uint64_t) * rows * n);
    for (size_t j = 0; j < rows; j++) /* :674-678 */
        ref_ntt_inverse(copy + j * n, &c->key_tables[j]);
    /* RNSBase::compose_array (rns.cpp:386-450): x = sum_i (x_i * (Q/q_i)^{-1} mod q_i) * (Q/q_i) mod Q */
    uint64_t Q[64], half[65], punct[64][64], invp[64];
    big_product(c, rows, Q);
    {
        /* upper_half_threshold = (Q + 1) >> 1 (context.cpp:370-376) */
        unsigned __int128 carry = 1;
        for (size_t l = 0; l < rows; l++)
        {
            carry += Q[l];
            half[l] = (uint64_t)carry;
            carry >>= 64;
        }
        half[rows] = (uint64_t)carry;
        for (size_t l = 0; l < rows; l++)
            half[l] = (half[l] >> 1) | (half[l + 1] << 63);
    }
    for (size_t i = 0; i < rows; i++)
    {
        const uint64_t qi = c->key_mod[i].value;
        big_divide_word(Q, rows, qi, punct[i]);
        uint64_t r = big_mod_word(punct[i], rows, qi);
        ref_try_invert_uint_mod(r, qi, &invp[i]);
    }
    double *res = (double *)calloc(2 * n, sizeof(double));
    const double two_pow_64 = 18446744073709551616.0;
    for (size_t ci = 0; ci < n; ci++)
    {
        uint64_t acc[65] = { 0 };
        for (size_t i = 0; i < rows; i++)
        {
            const uint64_t t = ref_multiply_uint_mod(copy[i * n + ci], invp[i], &c->key_mod[i]);
            unsigned __int128 carry = 0;
            for (size_t l = 0; l < rows; l++) /* acc += t * punct_i */
            {
                carry += (unsigned __int128)t * punct[i][l] + acc[l];
                acc[l] = (uint64_t)carry;
                carry >>= 64;
            }
            acc[rows] += (uint64_t)carry;
            /* acc < 2Q: one conditional subtraction (add_uint_uint_mod) */
            int ge = acc[rows] != 0;
            if (!ge)
            {
                ge = 1;
                for (size_t l = rows; l-- > 0;)
                    if (acc[l] != Q[l])
                    {
                        ge = acc[l] > Q[l];
                        break;
                    }
            }
            if (ge)
            {
                unsigned __int128 borrow = 0;
                for (size_t l = 0; l < rows; l++)
                {
                    const unsigned __int128 d = (unsigned __int128)acc[l] - Q[l] - borrow;
                    acc[l] = (uint64_t)d;
                    borrow = (d >> 64) & 1;
                }
                acc[rows] = 0;
            }
        }
        int upper = 1; /* is_greater_than_or_equal_uint(acc, upper_half_threshold) */
        for (size_t l = rows; l-- > 0;)
            if (acc[l] != half[l])
            {
                upper = acc[l] > half[l];
                break;
            }
        double r = 0.0, scaled = inv_scale; /* :686-715 */
        for (size_t j = 0; j < rows; j++, scaled *= two_pow_64)
        {
            if (upper)
            {
                if (acc[j] > Q[j])
                {
                    const uint64_t diff = acc[j] - Q[j];
                    r += diff ? (double)diff * scaled : 0.0;
                }
                else
                {
                    const uint64_t diff = Q[j] - acc[j];
                    r -= diff ? (double)diff * scaled : 0.0;
                }
            }
            else
                r += acc[j] ? (double)acc[j] * scaled : 0.0;
        }
        res[2 * ci] = r;
    }
    size_t tt = n;
    for (int i = 0; i < logn; i++) /* :723-741 */
    {
        const size_t mm = (size_t)1 << i;
        tt >>= 1;
        for (size_t j = 0; j < mm; j++)
        {
            const size_t j1 = 2 * j * tt;
            const double sr = enc->roots[2 * (mm + j)], si = enc->roots[2 * (mm + j) + 1];
            for (size_t k = j1; k < j1 + tt; k++)
            {
                const double ur = res[2 * k], ui = res[2 * k + 1], xr = res[2 * (k + tt)], xi = res[2 * (k + tt) + 1];
                const double vr = xr * sr - xi * si, vi = xr * si + xi * sr;
                res[2 * k] = ur + vr;
                res[2 * k + 1] = ui + vi;
                res[2 * (k + tt)] = ur - vr;
                res[2 * (k + tt) + 1] = ui - vi;
            }
        }
    }
    for (size_t i = 0; i < slots; i++) /* :743-746 */
    {
        values[2 * i] = res[2 * enc->index_map[i]];
        values[2 * i + 1] = res[2 * enc->index_map[i] + 1];
    }
    free(res);
    free(copy);
    return 0;
}

void ref_generate_kswitch_key(const ref_context *c, const uint64_t *sk_ntt, const uint64_t *new_key_ntt,
                              uint64_t *state, uint64_t *key)
{
    const size_t n = c->n, n_key = c->n_key, n_ct = c->k_first, nsp = c->nsp;
    const size_t digits = (n_ct + nsp - 1) / nsp;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * n);
    for (size_t j = 0; j < digits; j++)
    {
        uint64_t *dst = key + j * 2 * n_key * n;
        ref_encrypt_zero_symmetric(c, n_key, sk_ntt, 1, state, dst); /* keygenerator.cpp:347 */
        const size_t r0 = j * nsp, r1 = r0 + nsp < n_ct ? r0 + nsp : n_ct;
        for (size_t r = r0; r < r1; r++) /* :351-362 */
        {
            const ref_modulus *m = &c->key_mod[r];
            uint64_t factor = 1;
            for (size_t s = 0; s < nsp; s++)
                factor = ref_multiply_uint_mod(factor, c->key_mod[n_ct + s].value % m->value, m);
            ref_multiply_poly_scalar_coeffmod(new_key_ntt + r * n, n, factor, m, temp);
            ref_add_poly_coeffmod(dst + r * n, temp, n, m, dst + r * n);
        }
    }
    free(temp);
}

void ref_dot_product_ct_sk(const ref_context *c, size_t k, const uint64_t *ct, size_t size, int is_ntt_form,
                           const uint64_t *sk_powers, uint64_t *out)
{
    const size_t n = c->n, n_key = c->n_key;
    uint64_t *copy = (uint64_t *)malloc(sizeof(uint64_t) * (size - 1) * k * n);
    memcpy(copy, ct + k * n, sizeof(uint64_t) * (size - 1) * k * n); /* decryptor.cpp:237-238 */
    memset(out, 0, sizeof(uint64_t) * k * n);
    for (size_t i = 0; i + 1 < size; i++)
        for (size_t r = 0; r < k; r++)
        {
            uint64_t *row = copy + (i * k + r) * n;
            const ref_modulus *m = &c->key_mod[r];
            if (!is_ntt_form) /* :241-244 */
                ref_ntt_forward_lazy(row, &c->key_tables[r], c->mode == REF_MODE_STRICT);
            ref_dyadic_product_coeffmod(row, sk_powers + (i * n_key + r) * n, n, m, row); /* :247-250 */
            ref_add_poly_coeffmod(out + r * n, row, n, m, out + r * n);                    /* :253-256 */
        }
    for (size_t r = 0; r < k; r++)
    {
        if (!is_ntt_form) /* :258-262 */
            ref_ntt_inverse(out + r * n, &c->key_tables[r]);
        ref_add_poly_coeffmod(out + r * n, ct + r * n, n, &c->key_mod[r], out + r * n); /* :265 */
    }
    free(copy);
}

int ref_decrypt_scale_and_round(ref_context *c, size_t k, const uint64_t *in, uint64_t *out)
{
    const ref_rns_tool *rt = ref_context_rns_tool(c, k);
    if (!rt)
        return -1;
    const size_t n = c->n;
    const uint64_t t = rt->t.value, gamma = rt->gamma.value;
    uint64_t qv[64], tg[2] = { t, gamma };
    for (size_t i = 0; i < k; i++)
        qv[i] = rt->q[i].value;
    ref_base_converter conv; /* base_q_to_t_gamma_conv_, rns.cpp:634-638 */
    if (ref_base_converter_init(&conv, qv, k, tg, 2))
        return -1;
    uint64_t neg_inv_q[2], inv_gamma_mod_t;
    for (int i = 0; i < 2; i++) /* rns.cpp:707-716 */
    {
        uint64_t pq = 1 % tg[i];
        for (size_t j = 0; j < k; j++)
            pq = ref_multiply_uint_mod(pq, qv[j] % tg[i], &conv.obase[i]);
        if (!ref_try_invert_uint_mod(pq, tg[i], &neg_inv_q[i]))
            return -1;
        neg_inv_q[i] = neg_inv_q[i] ? tg[i] - neg_inv_q[i] : 0;
    }
    if (!ref_try_invert_uint_mod(gamma % t, t, &inv_gamma_mod_t)) /* :693 */
        return -1;
    uint64_t *temp = (uint64_t *)malloc(sizeof(uint64_t) * k * n);
    uint64_t *ttg = (uint64_t *)malloc(sizeof(uint64_t) * 2 * n);
    for (size_t i = 0; i < k; i++) /* :1076-1082, prod_t_gamma_mod_q_ (:699-704) */
    {
        const uint64_t ptg = ref_multiply_uint_mod(t % qv[i], gamma % qv[i], &rt->q[i]);
        ref_multiply_poly_scalar_coeffmod(in + i * n, n, ptg, &rt->q[i], temp + i * n);
    }
    ref_fast_convert_array(&conv, temp, n, ttg); /* :1088 */
    for (int i = 0; i < 2; i++)                  /* :1091-1096 */
        ref_multiply_poly_scalar_coeffmod(ttg + i * n, n, neg_inv_q[i], &conv.obase[i], ttg + i * n);
    const uint64_t gamma_div_2 = gamma >> 1;
    for (size_t i = 0; i < n; i++) /* :1104-1125 */
    {
        uint64_t d;
        if (ttg[n + i] > gamma_div_2)
        {
            const uint64_t a = ttg[i] + (gamma - ttg[n + i]) % t;
            d = a >= t ? a - t : a;
        }
        else
        {
            const uint64_t b = ttg[n + i] % t;
            d = ttg[i] >= b ? ttg[i] - b : ttg[i] + t - b;
        }
        if (d)
            d = ref_multiply_uint_mod(d, inv_gamma_mod_t, &conv.obase[0]);
        out[i] = d;
    }
    free(temp);
    free(ttg);
    ref_base_converter_free(&conv);
    return 0;
}

/* ------------------------------------------------------------------------------------------
 * Synthetic data helpers
 * ---------------------------------------------------------------------------------------- */
uint64_t ref_splitmix64(uint64_t *state)
{
    uint64_t z = (*state += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

uint64_t ref_fnv1a64(const uint64_t *words, size_t count)
{
    uint64_t h = 0xcbf29ce484222325ULL;
    for (size_t i = 0; i < count; i++)
        for (int b = 0; b < 8; b++)
        {
            h ^= (words[i] >> (8 * b)) & 0xff;
            h *= 0x100000001b3ULL;
        }
    return h;
}

void ref_fill_rows(uint64_t *dst, size_t rows, size_t n, const uint64_t *moduli, uint64_t *state)
{
    for (size_t i = 0; i < rows; i++)
        for (size_t j = 0; j < n; j++)
            dst[i * n + j] = ref_splitmix64(state) % moduli[i];
}

/* ------------------------------------------------------------------------------------------
 * seeded ciphertexts. Ciphertext::expand_seed (ciphertext.cpp:126-133) = sample_poly_uniform
 * (util/rlwe.cpp:101-129) driven by BlakePRNG (randomgen.h:199-222, randomgen.cpp:63-73), which fills 4096-byte buffers
 * with blake2xb(counter; key = seed) (util/blake2xb.c, vendored BLAKE2 reference code). BLAKE2b below is RFC 7693's
 * pseudocode written out; the parameter block and the XOF construction follow the BLAKE2 / BLAKE2X specifications.
 * Pinned by tests/golden/prng_vectors.json, generated with the reference's own blake2b.c / blake2xb.c (oracle/_ref).
 * ---------------------------------------------------------------------------------------- */
static const uint64_t b2_iv[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL, 0xa54ff53a5f1d36f1ULL,
                                   0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL, 0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
static const uint8_t b2_sigma[10][16] = {
    { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
    { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
    { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
    { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
    { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 }
};
#define B2_ROTR(x, n) (((x) >> (n)) | ((x) << (64 - (n))))
/* F(h, m, t, f): RFC 7693 section 3.2; t is the 128-bit byte counter (only its low word is ever non-zero here) */
static void b2_F(uint64_t h[8], const uint8_t block[128], uint64_t t, int last)
{
    uint64_t m[16], v[16];
    for (int i = 0; i < 16; i++)
    {
        m[i] = 0;
        for (int b = 7; b >= 0; b--)
            m[i] = (m[i] << 8) | block[8 * i + b];
    }
    for (int i = 0; i < 8; i++)
    {
        v[i] = h[i];
        v[i + 8] = b2_iv[i];
    }
    v[12] ^= t;
    if (last)
        v[14] ^= 0xFFFFFFFFFFFFFFFFULL;
    static const int idx[8][4] = { { 0, 4, 8, 12 }, { 1, 5, 9, 13 }, { 2, 6, 10, 14 }, { 3, 7, 11, 15 },
                                   { 0, 5, 10, 15 }, { 1, 6, 11, 12 }, { 2, 7, 8, 13 }, { 3, 4, 9, 14 } };
    for (int r = 0; r < 12; r++)
    {
        const uint8_t *sg = b2_sigma[r % 10];
        for (int g = 0; g < 8; g++)
        {
            const int a = idx[g][0], b = idx[g][1], c = idx[g][2], d = idx[g][3];
            v[a] = v[a] + v[b] + m[sg[2 * g]];
            v[d] = B2_ROTR(v[d] ^ v[a], 32);
            v[c] = v[c] + v[d];
            v[b] = B2_ROTR(v[b] ^ v[c], 24);
            v[a] = v[a] + v[b] + m[sg[2 * g + 1]];
            v[d] = B2_ROTR(v[d] ^ v[a], 16);
            v[c] = v[c] + v[d];
            v[b] = B2_ROTR(v[b] ^ v[c], 63);
        }
    }
    for (int i = 0; i < 8; i++)
        h[i] ^= v[i] ^ v[i + 8];
}
/* one BLAKE2b instance over a message held in memory, with the full parameter block */
static void b2_hash(uint8_t *out, unsigned outlen, const uint8_t *msg, size_t msglen, const uint8_t *key, unsigned keylen,
                    unsigned fanout, unsigned depth, uint32_t leaf_length, uint32_t node_offset, uint32_t xof_length,
                    unsigned inner_length)
{
    uint8_t p[64] = { 0 };
    p[0] = (uint8_t)outlen;
    p[1] = (uint8_t)keylen;
    p[2] = (uint8_t)fanout;
    p[3] = (uint8_t)depth;
    for (int i = 0; i < 4; i++)
    {
        p[4 + i] = (uint8_t)(leaf_length >> (8 * i));
        p[8 + i] = (uint8_t)(node_offset >> (8 * i));
        p[12 + i] = (uint8_t)(xof_length >> (8 * i));
    }
    p[17] = (uint8_t)inner_length;
    uint64_t h[8];
    for (int i = 0; i < 8; i++)
    {
        uint64_t w = 0;
        for (int b = 7; b >= 0; b--)
            w = (w << 8) | p[8 * i + b];
        h[i] = b2_iv[i] ^ w;
    }
    /* the data to compress: the key padded to one block (if any), then the message; the last block is zero padded */
    const size_t total = (keylen ? 128 : 0) + msglen;
    uint8_t *data = (uint8_t *)calloc(total + 128, 1);
    if (keylen)
        memcpy(data, key, keylen);
    memcpy(data + (keylen ? 128 : 0), msg, msglen);
    const size_t nblocks = total ? (total + 127) / 128 : 1;
    for (size_t b = 0; b < nblocks; b++)
    {
        const int last = b + 1 == nblocks;
        b2_F(h, data + 128 * b, last ? total : 128 * (b + 1), last);
    }
    free(data);
    for (unsigned i = 0; i < outlen; i++)
        out[i] = (uint8_t)(h[i >> 3] >> (8 * (i & 7)));
}
/* BLAKE2Xb (blake2xb.c:36-187): root hash with the XOF length in its parameter block, then one 64-byte-input node per
 * output block */
int ref_blake2xb(uint8_t *out, size_t outlen, const uint8_t *in, size_t inlen, const uint8_t *key, size_t keylen)
{
    if (outlen == 0 || outlen > 0xFFFFFFFFULL || keylen > 64)
        return -1;
    uint8_t h0[64];
    b2_hash(h0, 64, in, inlen, key, (unsigned)keylen, 1, 1, 0, 0, (uint32_t)outlen, 0);
    size_t done = 0;
    for (uint32_t i = 0; done < outlen; i++)
    {
        const unsigned take = (unsigned)(outlen - done < 64 ? outlen - done : 64);
        b2_hash(out + done, take, h0, 64, NULL, 0, 0, 0, 64, i, (uint32_t)outlen, 64);
        done += take;
    }
    return 0;
}
/* expand_seed: rows x n words, row j uniform below moduli[j] (util/rlwe.cpp:101-129 over BlakePRNG) */
void ref_expand_seed(const uint64_t seed[8], const uint64_t *moduli, size_t rows, size_t n, uint64_t *dst)
{
    uint8_t key[64], buffer[4096];
    for (int i = 0; i < 8; i++)
        for (int b = 0; b < 8; b++)
            key[8 * i + b] = (uint8_t)(seed[i] >> (8 * b));
    uint64_t counter = 0;
    size_t head = sizeof(buffer);
    const uint64_t max_random = 0x7FFFFFFFFFFFFFFFULL;
    for (size_t j = 0; j < rows; j++)
    {
        const uint64_t q = moduli[j], max_multiple = max_random - max_random % q - 1;
        for (size_t i = 0; i < n; i++)
        {
            uint64_t r;
            do
            {
                uint32_t w[2];
                for (int t = 0; t < 2; t++)
                {
                    if (head == sizeof(buffer)) /* randomgen.cpp:63-73 */
                    {
                        uint8_t ctr[8];
                        for (int b = 0; b < 8; b++)
                            ctr[b] = (uint8_t)(counter >> (8 * b));
                        ref_blake2xb(buffer, sizeof(buffer), ctr, 8, key, 64);
                        counter++;
                        head = 0;
                    }
                    w[t] = (uint32_t)buffer[head] | ((uint32_t)buffer[head + 1] << 8) | ((uint32_t)buffer[head + 2] << 16) |
                           ((uint32_t)buffer[head + 3] << 24);
                    head += 4;
                }
                r = ((uint64_t)w[0] << 31) | ((uint64_t)w[1] >> 1); /* rlwe.cpp:124 */
            } while (r >= max_multiple);
            dst[j * n + i] = r % q;
        }
    }
}
