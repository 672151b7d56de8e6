// keyswitch.hip -- column-parallel kernels of the hybrid (multi-special-prime) key switch:
// modup_rns / modup_to_single_rns (native/src/seal/multi_special_primes.cpp:80-185), the 128-bit
// inner product against the key (native/src/seal/evaluator.cpp:2315-2357) and
// rescale_special_rns_inplace (multi_special_primes.cpp:237-304).
//
// The reference recomputes the punctured products and inverses on every call
// (multi_special_primes.cpp:110-126, :244-248, :292-299); here they are context constants (KsDev).
// The reference also keeps two (k+nsp) x N arrays of 128-bit accumulators in memory across the digit
// loop; here one lane owns one (row, coefficient) and keeps both accumulators in registers across
// all digits, streaming the key slices.
#include <cstdlib>
#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        constexpr int kThreads = 256;

        // modup constants of digit j: [inv_punch(nsp) | inv_punch_shoup(nsp) | punch[rows][nsp]]
        __device__ __forceinline__ const u64 *modup_block(const KsDev *d, int j)
        {
            const int rows = d->k + d->nsp;
            return d->modup + static_cast<std::size_t>(j) * (2 * d->nsp + rows * d->nsp);
        }

        // For every digit j (or only `only_digit`): read the bundle's coefficient-form rows and write every
        // row outside the bundle of ext[item][j].
        __global__ __launch_bounds__(kThreads) void ks_modup_kernel(const KsDev *__restrict__ d,
                                                                    const PrimeDev *__restrict__ primes,
                                                                    const u64 *__restrict__ coeff,
                                                                    std::size_t coeff_stride, u64 *__restrict__ ext,
                                                                    std::size_t ext_stride,
                                                                    std::size_t ext_digit_stride, std::size_t count,
                                                                    int logn, int only_digit)
        {
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t item = i >> logn;
            if (item >= count)
                return;
            const std::size_t c = i & ((static_cast<std::size_t>(1) << logn) - 1);
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const int k = d->k, nsp = d->nsp, rows = k + nsp;
            const u64 *src = coeff + item * coeff_stride + c;
            const int j_begin = only_digit >= 0 ? only_digit : 0;
            const int j_end = only_digit >= 0 ? only_digit + 1 : d->nd;
            for (int j = j_begin; j < j_end; j++)
            {
                const int r0 = j * nsp;
                const int r1 = r0 + nsp < k ? r0 + nsp : k;
                const int bs = r1 - r0;
                u64 *dst = ext + item * ext_stride + static_cast<std::size_t>(j) * ext_digit_stride + c;
                if (bs == 1)
                {
                    // multi_special_primes.cpp:99-108
                    const u64 x = src[r0 * N];
                    const u64 psrc = primes[d->row_prime[r0]].p;
                    for (int r = 0; r < rows; r++)
                    {
                        if (r == r0)
                            continue;
                        const PrimeDev &D = primes[d->row_prime[r]];
                        dst[r * N] = psrc <= D.p ? x : barrett_reduce_63(x, D.p, D.cr1);
                    }
                    continue;
                }
                const u64 *blk = modup_block(d, j);
                u64 y[kMaxModuli > 8 ? 8 : kMaxModuli]; // bundles wider than 8 fall back to recomputation
                const bool cached = bs <= 8;
                if (cached)
                {
#pragma unroll
                    for (int a = 0; a < 8; a++)
                        if (a < bs)
                        {
                            const u64 p = primes[d->row_prime[r0 + a]].p;
                            y[a] = mulmod_shoup(src[(r0 + a) * N], blk[a], blk[nsp + a], p); // :135-139
                        }
                }
                for (int r = 0; r < rows; r++)
                {
                    if (r >= r0 && r < r1)
                        continue;
                    const PrimeDev &D = primes[d->row_prime[r]];
                    const u64 *punch = blk + 2 * nsp + r * nsp;
                    u64 lo = 0, hi = 0;
                    if (cached)
                    {
#pragma unroll
                        for (int a = 0; a < 8; a++)
                            if (a < bs)
                                mac128(lo, hi, y[a], punch[a]);
                    }
                    else
                    {
                        for (int a = 0; a < bs; a++)
                        {
                            const u64 p = primes[d->row_prime[r0 + a]].p;
                            mac128(lo, hi, mulmod_shoup(src[(r0 + a) * N], blk[a], blk[nsp + a], p), punch[a]);
                        }
                    }
                    dst[r * N] = barrett_reduce_128(lo, hi, D.p, D.cr0, D.cr1); // :143-146
                }
            }
        }

        // prod[item][l][r][c] = barrett_reduce_128( sum_j ct_j[r][c] * key[j][l][row_prime[r]][c] )
        // where ct_j = the (NTT-form) target row if r is in bundle j, else ext[j][item][r] (digit-major)
        // (evaluator.cpp:2315-2349). One lane per (item, row, coefficient).
        __global__ __launch_bounds__(kThreads) void ks_mac_kernel(const KsDev *__restrict__ d,
                                                                  const PrimeDev *__restrict__ primes,
                                                                  const u64 *__restrict__ target,
                                                                  std::size_t target_stride,
                                                                  const u64 *__restrict__ ext, std::size_t ext_stride,
                                                                  std::size_t ext_digit_stride,
                                                                  const u64 *__restrict__ key,
                                                                  u64 *__restrict__ prod, std::size_t prod_stride,
                                                                  std::size_t count, int logn, int j0, int j1)
        {
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const int k = d->k, nsp = d->nsp, rows = k + nsp, n_total = d->n_total;
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t c = i & (N - 1);
            const std::size_t rr = i >> logn;
            const int r = static_cast<int>(rr % rows);
            const std::size_t item = rr / rows;
            if (item >= count)
                return;
            const int rns_idx = d->row_prime[r];
            const int my_digit = r < k ? r / nsp : -1;
            const u64 *pext = ext + item * ext_stride + static_cast<std::size_t>(r) * N + c;
            const u64 *pkey = key + static_cast<std::size_t>(rns_idx) * N + c;
            const std::size_t key_comp = static_cast<std::size_t>(n_total) * N;
            u64 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
            for (int j = j0; j < j1; j++) // (all digits of the level, or one device's share of them: latency mode)
            {
                const u64 x = j == my_digit ? target[item * target_stride + static_cast<std::size_t>(r) * N + c]
                                            : pext[static_cast<std::size_t>(j) * ext_digit_stride];
                const u64 k0 = pkey[(2 * static_cast<std::size_t>(j)) * key_comp];
                const u64 k1 = pkey[(2 * static_cast<std::size_t>(j) + 1) * key_comp];
                mac128(lo0, hi0, x, k0);
                mac128(lo1, hi1, x, k1);
            }
            const PrimeDev &P = primes[rns_idx];
            u64 *pp = prod + item * prod_stride + static_cast<std::size_t>(r) * N + c;
            pp[0] = barrett_reduce_128(lo0, hi0, P.p, P.cr0, P.cr1);
            pp[static_cast<std::size_t>(rows) * N] = barrett_reduce_128(lo1, hi1, P.p, P.cr0, P.cr1);
        }

        // The same inner product with the key words of a lane's (row, coefficient) kept in registers across
        // `group` consecutive ciphertexts: the key slice (2 * nd words per lane, 29 MB per pass at cfg3) is read once
        // per group instead of once per ciphertext (it made up two thirds of this kernel's read traffic at group 1, and
        // still 17 % at group 8: config 4 reads 132 digit rows, writes 24 and re-reads 264 / group key rows per
        // ciphertext). The digit words of the next ciphertext are requested before the current one is accumulated, so the
        // group size costs no registers; the launcher picks it from the batch (mac_group).
        template <int ND>
        __global__ __launch_bounds__(kThreads) void ks_mac_items_kernel(const KsDev *__restrict__ d,
                                                                        const PrimeDev *__restrict__ primes,
                                                                        const u64 *__restrict__ target,
                                                                        std::size_t target_stride,
                                                                        const u64 *__restrict__ ext,
                                                                        std::size_t ext_stride,
                                                                        std::size_t ext_digit_stride,
                                                                        const u64 *__restrict__ key,
                                                                        u64 *__restrict__ prod, std::size_t prod_stride,
                                                                        std::size_t count, int logn, std::size_t group)
        {
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const int k = d->k, nsp = d->nsp, rows = k + nsp, n_total = d->n_total;
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t c = i & (N - 1);
            const std::size_t rr = i >> logn;
            const int r = static_cast<int>(rr % rows);
            const std::size_t item0 = (rr / rows) * group;
            if (item0 >= count)
                return;
            const std::size_t item1 = item0 + group < count ? item0 + group : count;
            const int rns_idx = d->row_prime[r];
            const int my_digit = r < k ? r / nsp : -1;
            const std::size_t row_off = static_cast<std::size_t>(r) * N + c;
            const u64 *pkey = key + static_cast<std::size_t>(rns_idx) * N + c;
            const std::size_t key_comp = static_cast<std::size_t>(n_total) * N;
            u64 k0[ND], k1[ND], x[ND], xn[ND];
#pragma unroll
            for (int j = 0; j < ND; j++)
            {
                k0[j] = pkey[(2 * static_cast<std::size_t>(j)) * key_comp];
                k1[j] = pkey[(2 * static_cast<std::size_t>(j) + 1) * key_comp];
            }
            auto load_x = [&](u64(&dst)[ND], std::size_t item) {
#pragma unroll
                for (int j = 0; j < ND; j++)
                    dst[j] = j == my_digit ? target[item * target_stride + row_off]
                                           : ext[item * ext_stride + static_cast<std::size_t>(j) * ext_digit_stride + row_off];
            };
            load_x(x, item0);
            const PrimeDev &P = primes[rns_idx];
            const u64 p = P.p, cr0 = P.cr0, cr1 = P.cr1;
            for (std::size_t item = item0; item < item1; item++)
            {
                if (item + 1 < item1)
                    load_x(xn, item + 1);
                u64 lo0 = 0, hi0 = 0, lo1 = 0, hi1 = 0;
#pragma unroll
                for (int j = 0; j < ND; j++)
                {
                    mac128(lo0, hi0, x[j], k0[j]);
                    mac128(lo1, hi1, x[j], k1[j]);
                }
                u64 *pp = prod + item * prod_stride + row_off;
                store_stream(pp, barrett_reduce_128(lo0, hi0, p, cr0, cr1));
                store_stream(pp + static_cast<std::size_t>(rows) * N, barrett_reduce_128(lo1, hi1, p, cr0, cr1));
#pragma unroll
                for (int j = 0; j < ND; j++)
                    x[j] = xn[j];
            }
        }

        // rescale_special_rns_inplace, steps 1-2 (multi_special_primes.cpp:253-282): from the (coefficient
        // form) special rows of one polynomial compute temp_i for every ciphertext prime i.
        __global__ __launch_bounds__(kThreads) void ks_moddown_pre_kernel(const KsDev *__restrict__ d,
                                                                          const PrimeDev *__restrict__ primes,
                                                                          const u64 *__restrict__ prod,
                                                                          std::size_t prod_stride,
                                                                          u64 *__restrict__ temp,
                                                                          std::size_t temp_stride, std::size_t npolys,
                                                                          int logn)
        {
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t poly = i >> logn;
            if (poly >= npolys)
                return;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const std::size_t c = i & (N - 1);
            const int k = d->k, nsp = d->nsp;
            const u64 *sp = prod + poly * prod_stride + static_cast<std::size_t>(k) * N + c;
            u64 *pt = temp + poly * temp_stride + c;
            if (nsp == 1)
            {
                const PrimeDev &S = primes[d->row_prime[k]];
                const u64 v = neg_mod(barrett_reduce_63(sp[0], S.p, S.cr1), S.p); // :270-273
                for (int q = 0; q < k; q++)
                {
                    const PrimeDev &Q = primes[d->row_prime[q]];
                    pt[q * N] = barrett_reduce_63(v, Q.p, Q.cr1); // (v < P < 2^61: the one-word Barrett step gives the same canonical residue)
                }
                return;
            }
            for (int q = 0; q < k; q++)
            {
                const PrimeDev &Q = primes[d->row_prime[q]];
                u64 lo = 0, hi = 0;
                for (int j = 0; j < nsp; j++)
                {
                    const u64 pj = primes[d->row_prime[k + j]].p;
                    const u64 y = mulmod_shoup(sp[j * N], d->inv_hat[j], d->inv_hat_shoup[j], pj); // :262-267
                    mac128(lo, hi, y, d->neg_hat[q * nsp + j]);
                }
                pt[q * N] = barrett_reduce_128(lo, hi, Q.p, Q.cr0, Q.cr1);
            }
        }

        // step 4 (multi_special_primes.cpp:291-302) + the final add_poly_coeffmod of evaluator.cpp:2363-2366
        __global__ __launch_bounds__(kThreads) void ks_moddown_post_kernel(const KsDev *__restrict__ d,
                                                                           const PrimeDev *__restrict__ primes,
                                                                           u64 *__restrict__ prod,
                                                                           std::size_t prod_stride,
                                                                           const u64 *__restrict__ temp,
                                                                           std::size_t temp_stride,
                                                                           u64 *__restrict__ ct,
                                                                           std::size_t ct_item_stride,
                                                                           std::size_t npolys, int logn,
                                                                           int add_into_ct, const u64 *__restrict__ c0_src,
                                                                           std::size_t c0_stride, unsigned *__restrict__ tflags)
        {
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const int k = d->k;
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t c = i & (N - 1);
            const std::size_t rr = i >> logn;
            const int q = static_cast<int>(rr % k);
            const std::size_t poly = rr / k;
            if (poly >= npolys)
                return;
            const PrimeDev &Q = primes[d->row_prime[q]];
            u64 *pp = prod + poly * prod_stride + static_cast<std::size_t>(q) * N + c;
            const u64 v = mulmod_shoup(*pp + temp[poly * temp_stride + static_cast<std::size_t>(q) * N + c], d->invP[q],
                                       d->invP_shoup[q], Q.p);
            if (add_into_ct)
            {
                // polynomial `poly` is component (poly & 1) of ciphertext (poly >> 1)
                u64 *pc = ct + (poly >> 1) * ct_item_stride + ((poly & 1) * static_cast<std::size_t>(k) + q) * N + c;
                u64 w;
                if (!c0_src)
                    w = add_mod(v, *pc, Q.p);
                else // apply_galois: the ciphertext is (c0_src, 0) and only written here (evaluator.cpp:1903-1935)
                    w = (poly & 1) ? v : add_mod(v, c0_src[(poly >> 1) * c0_stride + static_cast<std::size_t>(q) * N + c], Q.p);
                *pc = w;
                if (poly & 1)
                    note_nonzero(tflags, poly >> 1, w);
            }
            else
                *pp = v;
        }

        // BFV: rescale_special_rns_inplace (multi_special_primes.cpp:237-304) + the final add (evaluator.cpp:2363-2366)
        // in one kernel. In BFV nothing happens between steps 2 and 4 of the rescale (the q rows are simply brought
        // to coefficient form), so temp_q is recomputed per lane from the special rows instead of being written
        // by one kernel and read by the next (2*k row transfers per polynomial), and with DEFER the top inverse-NTT
        // layer of every row read here is applied on load (kNttDeferTop: no ntt_inv_top pass over the 2*(k+nsp) rows).
        // Every value that leaves this kernel is a canonical residue of an exact modular expression, so the
        // representatives chosen for the intermediates do not matter.
        // both coefficients c and c + N/2 of one row: with DEFER the top inverse-NTT layer (BackwardLazyLast,
        // ntt.cpp:274-281) is applied to the pair, else the two words are used as they are
        template <bool DEFER>
        __device__ __forceinline__ void row_pair(const u64 *__restrict__ row, std::size_t c_lo, std::size_t half,
                                                 const PrimeDev &P, u64 &x_lo, u64 &x_hi)
        {
            const u64 u = row[c_lo], v = row[c_lo + half];
            if (!DEFER)
            {
                x_lo = u;
                x_hi = v;
                return;
            }
            u64 tt = u + v;
            tt = tt >= P.two_p ? tt - P.two_p : tt;
            x_lo = mulmod_lazy(tt, P.inv_n, P.inv_n_shoup, P.p);
            x_hi = mulmod_lazy(u - v + P.two_p, P.inv_n_w, P.inv_n_w_shoup, P.p);
        }

        // One lane per (polynomial, coefficient pair c / c + N/2), walking the k ciphertext primes: the special rows (and, with
        // DEFER, their top inverse-NTT layer) are read and reduced ONCE per column pair -- round 3; one lane per (polynomial,
        // prime, pair) had every q lane fetch and transform them again: 6 of 28 row passes per polynomial at config 3 -- every
        // other input word exactly once.
        // ONE: a single special prime (every BASELINE config): its reduced pair lives in two registers across the loop; several
        // special primes re-read their rows per ciphertext prime (the lane's own addresses: L1 hits) as before.
        template <bool DEFER, bool ONE>
        __global__ __launch_bounds__(kThreads) void ks_moddown_bfv_kernel(const KsDev *__restrict__ d,
                                                                          const PrimeDev *__restrict__ primes,
                                                                          const u64 *__restrict__ prod,
                                                                          std::size_t prod_stride, u64 *__restrict__ ct,
                                                                          std::size_t ct_item_stride, std::size_t npolys,
                                                                          int logn, const u64 *__restrict__ c0_src,
                                                                          std::size_t c0_stride, unsigned *__restrict__ tflags)
        {
            const std::size_t N = static_cast<std::size_t>(1) << logn, half = N >> 1;
            const int k = d->k, nsp = d->nsp;
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            const std::size_t c = i & (half - 1);
            const std::size_t poly = i >> (logn - 1);
            if (poly >= npolys)
                return;
            const u64 *pp = prod + poly * prod_stride;
            // steps 1-2, the part that does not depend on the ciphertext prime (multi_special_primes.cpp:253-273)
            u64 sred[2] = {0, 0};
            if constexpr (ONE)
            {
                const PrimeDev &S = primes[d->row_prime[k]];
                u64 sv[2];
                row_pair<DEFER>(pp + static_cast<std::size_t>(k) * N, c, half, S, sv[0], sv[1]);
#pragma unroll
                for (int h = 0; h < 2; h++)
                    sred[h] = neg_mod(barrett_reduce_63(sv[h], S.p, S.cr1), S.p); // :270-273
            }
            u64 nz = 0;
            for (int q = 0; q < k; q++)
            {
                const PrimeDev &Q = primes[d->row_prime[q]];
                u64 temp[2];
                if constexpr (ONE)
                {
#pragma unroll
                    for (int h = 0; h < 2; h++)
                        // (sred < P < 2^61 is one word: the one-word Barrett step -- a 64-bit high product, a multiply, a
                        //  conditional subtraction -- gives the canonical residue the two-word form gave with four times the work)
                        temp[h] = barrett_reduce_63(sred[h], Q.p, Q.cr1);
                }
                else
                {
                    u64 lo[2] = {0, 0}, hi[2] = {0, 0};
                    for (int j = 0; j < nsp; j++)
                    {
                        const PrimeDev &S = primes[d->row_prime[k + j]];
                        u64 sv[2];
                        row_pair<DEFER>(pp + static_cast<std::size_t>(k + j) * N, c, half, S, sv[0], sv[1]);
#pragma unroll
                        for (int h = 0; h < 2; h++)
                        {
                            const u64 y = mulmod_shoup(sv[h], d->inv_hat[j], d->inv_hat_shoup[j], S.p); // :262-267
                            mac128(lo[h], hi[h], y, d->neg_hat[q * nsp + j]);
                        }
                    }
                    temp[0] = barrett_reduce_128(lo[0], hi[0], Q.p, Q.cr0, Q.cr1);
                    temp[1] = barrett_reduce_128(lo[1], hi[1], Q.p, Q.cr0, Q.cr1);
                }
                // step 4 (:291-302) and the add into the ciphertext
                u64 pv[2];
                row_pair<DEFER>(pp + static_cast<std::size_t>(q) * N, c, half, Q, pv[0], pv[1]);
                u64 *pc = ct + (poly >> 1) * ct_item_stride + ((poly & 1) * static_cast<std::size_t>(k) + q) * N + c;
#pragma unroll
                for (int h = 0; h < 2; h++)
                {
                    const u64 v = mulmod_shoup(pv[h] + temp[h], d->invP[q], d->invP_shoup[q], Q.p);
                    u64 w;
                    if (!c0_src)
                        w = add_mod(v, pc[h * half], Q.p);
                    else // apply_galois: the ciphertext is (c0_src, 0) and only written here (evaluator.cpp:1903-1935)
                        w = (poly & 1) ? v
                                       : add_mod(v, c0_src[(poly >> 1) * c0_stride + static_cast<std::size_t>(q) * N + c + h * half], Q.p);
                    pc[h * half] = w;
                    nz |= w;
                }
            }
            if (poly & 1) // component 1 of ciphertext poly >> 1: what is_transparent looks at
                note_nonzero(tflags, poly >> 1, nz);
        }

        inline unsigned blocks_for(std::size_t lanes)
        {
            return static_cast<unsigned>((lanes + kThreads - 1) / kThreads);
        }
    } // namespace

    hipError_t launch_ks_modup(const Engine &e, const KsDev *d, const KsDev &, const u64 *coeff,
                               std::size_t coeff_stride, u64 *ext, std::size_t ext_stride,
                               std::size_t ext_digit_stride, std::size_t count, int only_digit)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "ks_modup", 0);
        ks_modup_kernel<<<blocks_for(count << e.logn), kThreads, 0, e.lane().stream>>>(
            d, e.d_primes, coeff, coeff_stride, ext, ext_stride, ext_digit_stride, count, e.logn, only_digit);
        return hipGetLastError();
    }

    hipError_t launch_ks_mac(const Engine &e, const KsDev *d, const KsDev &h, const u64 *target,
                             std::size_t target_stride, const u64 *ext, std::size_t ext_stride,
                             std::size_t ext_digit_stride, const u64 *key, u64 *prod, std::size_t prod_stride,
                             std::size_t count, int j0, int j1)
    {
        if (!count)
            return hipSuccess;
        if (j1 < 0)
            j1 = h.nd;
        if (j0 < 0 || j0 > j1 || j1 > h.nd)
            return hipErrorInvalidValue;
        const bool all_digits = j0 == 0 && j1 == h.nd;
        const u64 *tg = target;
        const std::size_t lanes = (count * static_cast<std::size_t>(h.k + h.nsp)) << e.logn;
        ProfScope prof(e, "ks_mac", 0);
        // ciphertexts per key load: 8 for small batches (enough workgroups to fill the chip), else 16 -- or 64 when the key
        // slices this kernel reads (2 * nd * rows rows) are too large to stay in the memory-side cache between groups
        // (profiles/r02/ks_mac_group.txt: config 4, 69 MB of key: 9.15 -> 7.96 ms per step; config 5, 251 MB: 9.38 -> 7.66;
        // config 3, 29 MB: 3.68 -> 3.60 at 16, but 3.9-4.1 at 32)
        static const std::size_t forced = [] {
            const char *env = exp_env("SEALHIP_KS_MAC_GROUP"); // (measurement-only build)
            return env ? static_cast<std::size_t>(std::strtoull(env, nullptr, 10)) : std::size_t(0);
        }();
        const std::size_t key_bytes = (2ull * h.nd * (h.k + h.nsp)) << (e.logn + 3);
        // (the largest candidate that still leaves 4096 workgroups -- two full rounds of the chip's resident set)
        const std::size_t cap = key_bytes > (std::size_t(48) << 20) ? 64 : 16;
        std::size_t mac_group = 8;
        for (std::size_t g = cap; g > 8; g >>= 1)
            if ((((count + g - 1) / g * static_cast<std::size_t>(h.k + h.nsp)) << e.logn) / kThreads >= 4096)
            {
                mac_group = g;
                break;
            }
        if (forced)
            mac_group = forced;
        const std::size_t groups = (count + mac_group - 1) / mac_group;
        const std::size_t glanes = (groups * static_cast<std::size_t>(h.k + h.nsp)) << e.logn;
#define SEALHIP_KS_MAC(ND)                                                                                          \
    case ND:                                                                                                        \
        ks_mac_items_kernel<ND><<<blocks_for(glanes), kThreads, 0, e.lane().stream>>>(                                     \
            d, e.d_primes, tg, target_stride, ext, ext_stride, ext_digit_stride, key, prod, prod_stride, count,    \
            e.logn, mac_group);                                                                                     \
        break;
        switch (count >= 16 && all_digits ? h.nd : 0)
        {
            SEALHIP_KS_MAC(1)
            SEALHIP_KS_MAC(2)
            SEALHIP_KS_MAC(3)
            SEALHIP_KS_MAC(4)
            SEALHIP_KS_MAC(5)
            SEALHIP_KS_MAC(6)
            SEALHIP_KS_MAC(7)
            SEALHIP_KS_MAC(8)
            SEALHIP_KS_MAC(9)
            SEALHIP_KS_MAC(10)
            SEALHIP_KS_MAC(11)
            SEALHIP_KS_MAC(12)
            SEALHIP_KS_MAC(13)
            SEALHIP_KS_MAC(14)
            SEALHIP_KS_MAC(15)
            SEALHIP_KS_MAC(16)
        default: // small batches or more than 16 digits: one lane per (ciphertext, row, coefficient)
            ks_mac_kernel<<<blocks_for(lanes), kThreads, 0, e.lane().stream>>>(d, e.d_primes, tg, target_stride, ext, ext_stride,
                                                                        ext_digit_stride, key, prod, prod_stride, count,
                                                                        e.logn, j0, j1);
        }
#undef SEALHIP_KS_MAC
        return hipGetLastError();
    }

    hipError_t launch_ks_moddown_pre(const Engine &e, const KsDev *d, const KsDev &, const u64 *prod,
                                     std::size_t prod_stride, u64 *temp, std::size_t temp_stride, std::size_t npolys)
    {
        if (!npolys)
            return hipSuccess;
        ProfScope prof(e, "ks_moddown_pre", 0);
        ks_moddown_pre_kernel<<<blocks_for(npolys << e.logn), kThreads, 0, e.lane().stream>>>(
            d, e.d_primes, prod, prod_stride, temp, temp_stride, npolys, e.logn);
        return hipGetLastError();
    }

    hipError_t launch_ks_moddown_bfv(const Engine &e, const KsDev *d, const KsDev &h, const u64 *prod,
                                     std::size_t prod_stride, u64 *ct, std::size_t ct_item_stride, std::size_t npolys,
                                     bool top_deferred, const u64 *c0_src, std::size_t c0_stride)
    {
        if (!npolys)
            return hipSuccess;
        const std::size_t lanes = npolys << (e.logn - 1); // one lane per (polynomial, coefficient pair)
        ProfScope prof(e, "ks_moddown_bfv", 0);
#define SEALHIP_MODDOWN_BFV(DEFER_, ONE_)                                                                            \
    ks_moddown_bfv_kernel<DEFER_, ONE_><<<blocks_for(lanes), kThreads, 0, e.lane().stream>>>(                              \
        d, e.d_primes, prod, prod_stride, ct, ct_item_stride, npolys, e.logn, c0_src, c0_stride, e.lane().tsink_arm)
        if (top_deferred && h.nsp == 1)
            SEALHIP_MODDOWN_BFV(true, true);
        else if (top_deferred)
            SEALHIP_MODDOWN_BFV(true, false);
        else if (h.nsp == 1)
            SEALHIP_MODDOWN_BFV(false, true);
        else
            SEALHIP_MODDOWN_BFV(false, false);
#undef SEALHIP_MODDOWN_BFV
        return hipGetLastError();
    }

    hipError_t launch_ks_moddown_post(const Engine &e, const KsDev *d, const KsDev &h, u64 *prod,
                                      std::size_t prod_stride, const u64 *temp, std::size_t temp_stride, u64 *ct,
                                      std::size_t ct_item_stride, std::size_t npolys, int add_into_ct, const u64 *c0_src,
                                      std::size_t c0_stride)
    {
        if (!npolys)
            return hipSuccess;
        const std::size_t lanes = (npolys * static_cast<std::size_t>(h.k)) << e.logn;
        ProfScope prof(e, "ks_moddown_post", 0);
        ks_moddown_post_kernel<<<blocks_for(lanes), kThreads, 0, e.lane().stream>>>(
            d, e.d_primes, prod, prod_stride, temp, temp_stride, ct, ct_item_stride, npolys, e.logn, add_into_ct, c0_src, c0_stride,
            add_into_ct ? e.lane().tsink_arm : nullptr);
        return hipGetLastError();
    }
} // namespace sealhip
