// rns.hip -- BEHZ base conversions and modulus switching (native/src/seal/util/rns.cpp:731-1068) as
// column-parallel kernels: one lane owns one coefficient column of one item and walks the RNS rows,
// so every global access is a coalesced row segment; the per-prime constants are wave-uniform
// (scalar loads). All outputs are canonical residues, so constants are fused where the reference
// multiplies twice (e.g. m_tilde * q^_i^{-1}).
#include <type_traits>

#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        // compile-time loop: f(std::integral_constant<int, I>{}) for I = 0 .. N-1
        template <int N, int I = 0, class F>
        __device__ __forceinline__ void static_for(F &&f)
        {
            if constexpr (I < N)
            {
                f(std::integral_constant<int, I>{});
                static_for<N, I + 1>(f);
            }
        }

        constexpr int kThreads = 256;

        // The level constants never change after the context is built. Reading them through the constant address
        // space tells the compiler so: scalar loads that can be merged and hoisted above the kernel's own stores
        // (through a plain pointer every constant is re-fetched, and waited for, right before its use).
        template <class T>
        __device__ __forceinline__ const __attribute__((address_space(4))) T *kc(const T *p)
        {
            return (const __attribute__((address_space(4))) T *)p;
        }

        struct Cols
        {
            std::size_t item, c;
        };
        __device__ __forceinline__ bool column(std::size_t count, int logn, Cols &out)
        {
            const std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
            out.item = i >> logn;
            out.c = i & ((static_cast<std::size_t>(1) << logn) - 1);
            return out.item < count;
        }

        // t[i] = x_i * (q^_i)^{-1} mod q_i  (first loop of fast_convert_array, rns.cpp:476-485)
        // then out_j = sum_i t[i] * M[j][i] mod p_j  (dot_product_mod, rns.cpp:487-495)
        template <int KMAX>
        __device__ __forceinline__ u64 dot_mod(const u64 (&t)[KMAX], int k, const u64 *__restrict__ mrow,
                                               const PrimeDev &P)
        {
            u64 lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                    mac128(lo, hi, t[i], mrow[i]);
            return barrett_reduce_128(lo, hi, P.p, P.cr0, P.cr1);
        }

        // fastbconv_m_tilde (rns.cpp:1025-1068): k rows -> |Bsk|+1 rows
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void fastbconv_m_tilde_kernel(
            const RnsDev *__restrict__ d, const PrimeDev *__restrict__ primes, const u64 *__restrict__ in,
            std::size_t in_stride, u64 *__restrict__ out, std::size_t out_stride, std::size_t count, int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k, nB = d->nB;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            u64 t[KMAX];
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                {
                    const PrimeDev &Q = primes[d->q_prime[i]];
                    t[i] = mul_mod(pin[i * N], d->q_mt_inv[i], Q.p, Q.cr0, Q.cr1);
                }
            for (int j = 0; j < nB; j++)
                pout[j * N] = dot_mod<KMAX>(t, k, d->q_to_Bsk + j * k, primes[d->bsk_prime[j]]);
            u64 acc = 0; // modulus 2^32: only the low word matters
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                    acc += t[i] * d->q_to_mt[i];
            pout[nB * N] = acc & 0xFFFFFFFFull;
        }

        // sm_mrq (rns.cpp:925-981)
        __device__ __forceinline__ u64 sm_mrq_one(u64 in_b, u64 r_mt, const RnsDev *d, int j, const PrimeDev &Bp)
        {
            u64 temp = r_mt;
            if (temp >= (1ull << 31))
                temp += Bp.p - (1ull << 32);
            return mul_mod(mul_add_mod(d->prod_q_mod_Bsk[j], temp, in_b, Bp.p, Bp.cr0, Bp.cr1), d->inv_mt_mod_Bsk[j],
                           Bp.p, Bp.cr0, Bp.cr1);
        }
        template <class DP>
        __device__ __forceinline__ u64 r_m_tilde(u64 in_mt, DP d)
        {
            const u64 temp = (in_mt * d->inv_prod_q_mod_mt) & 0xFFFFFFFFull;
            return temp ? (1ull << 32) - temp : 0;
        }

        __global__ __launch_bounds__(kThreads) void sm_mrq_kernel(const RnsDev *__restrict__ d,
                                                                  const PrimeDev *__restrict__ primes,
                                                                  const u64 *__restrict__ in, std::size_t in_stride,
                                                                  u64 *__restrict__ out, std::size_t out_stride,
                                                                  std::size_t count, int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int nB = d->nB;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            const u64 r_mt = r_m_tilde(pin[nB * N], d);
            for (int j = 0; j < nB; j++)
                pout[j * N] = sm_mrq_one(pin[j * N], r_mt, d, j, primes[d->bsk_prime[j]]);
        }

        // fused fastbconv_m_tilde + sm_mrq (evaluator.cpp:345-348): k rows -> |Bsk| rows
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void bfv_lift_kernel(const RnsDev *__restrict__ d,
                                                                    const PrimeDev *__restrict__ primes,
                                                                    const u64 *__restrict__ in, std::size_t in_stride,
                                                                    u64 *__restrict__ out, std::size_t out_stride,
                                                                    std::size_t count, int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k, nB = d->nB;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            u64 t[KMAX];
            u64 acc = 0;
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                {
                    const PrimeDev &Q = primes[d->q_prime[i]];
                    t[i] = mul_mod(pin[i * N], d->q_mt_inv[i], Q.p, Q.cr0, Q.cr1);
                    acc += t[i] * d->q_to_mt[i];
                }
            const u64 r_mt = r_m_tilde(acc & 0xFFFFFFFFull, d);
            for (int j = 0; j < nB; j++)
            {
                const PrimeDev &Bp = primes[d->bsk_prime[j]];
                const u64 conv = dot_mod<KMAX>(t, k, d->q_to_Bsk + j * k, Bp);
                pout[j * N] = sm_mrq_one(conv, r_mt, d, j, Bp);
            }
        }

        // Constant-folded variant of bfv_lift_kernel (k <= 32): the chain
        //   conv_j = sum_i t_i M_ji mod b;  u = (prod_q*temp + conv_j) mod b;  out = u * m_tilde^{-1} mod b
        // (rns.cpp:1062-1067 + :960-980) is a composition of exact modular operations, so it equals
        //   out = ( sum_i t_i*(M_ji*m_tilde^{-1}) + temp*(prod_q*m_tilde^{-1}) ) mod b
        // with ONE 128-bit accumulation and ONE Barrett reduction per output instead of three.
        // KMAX < 0 means "exactly K = -KMAX primes, known at compile time" (no per-iteration guards, constants
        // fetched with wide scalar loads); KMAX > 0 is the guarded form for any k <= KMAX
        // TOP (exact-K instances): one lane per column PAIR (c, c + N/2); after a row's two outputs are formed the lane
        // applies the forward NTT's top layer to them (ForwardLazy, ntt.cpp:245-252, the very function the NTT kernel's load
        // phase would run on these two words) and stores the results. The transform that follows is launched with
        // kNttTopDone: its workgroups then read their own half only, and the layer's products are computed once per pair
        // instead of once per workgroup.
        template <int KMAX, bool TOP = false>
        __global__ __launch_bounds__(kThreads) void bfv_lift2_kernel(const RnsDev *__restrict__ d_,
                                                                     const PrimeDev *__restrict__ primes_,
                                                                     const u64 *__restrict__ in, std::size_t in_stride,
                                                                     u64 *__restrict__ out, std::size_t out_stride,
                                                                     std::size_t count, int logn)
        {
            static_assert(!TOP || KMAX < 0, "the paired-column form exists for the exact-K instances");
            Cols cc;
            if (!column(count, TOP ? logn - 1 : logn, cc))
                return;
            constexpr int KA = KMAX < 0 ? -KMAX : KMAX; // array extent
            constexpr int NC = TOP ? 2 : 1;             // columns per lane
            const auto *d = kc(d_);           // read-only for the lifetime of the context: constant address space
            const auto *primes = kc(primes_); // (scalar loads the compiler may merge and hoist)
            const int k = KMAX < 0 ? KA : d->k, nB = d->nB;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            u64 t[NC][KA];
            u64 acc[NC] = {};
            (void)primes;
#pragma unroll
            for (int i = 0; i < KA; i++)
                if (KMAX < 0 || i < k)
                {
                    const u64 qp = d->q_p[i];
#pragma unroll
                    for (int h = 0; h < NC; h++)
                    {
                        t[h][i] = mulmod_shoup_hs(pin[i * N + h * (N >> 1)], d->q_mt_inv[i], d->q_mt_inv_s[i], qp); // exact canonical product
                        acc[h] += t[h][i] * d->q_to_mt[i];
                    }
                }
            u64 r_mt[NC];
#pragma unroll
            for (int h = 0; h < NC; h++)
                r_mt[h] = r_m_tilde(acc[h] & 0xFFFFFFFFull, d);
            // exact-K instances are only launched when the host proved every REDC lands below 2p (RnsDev::redc_small):
            // a compile-time fact there, so the row loops carry no branch
            const bool small = KMAX < 0 ? true : d->redc_small != 0;
            SplitT ts[NC][KA];
            if constexpr (KMAX < 0)
                static_for<KA>([&](auto I) {
#pragma unroll
                    for (int h = 0; h < NC; h++)
                        ts[h][I.value] = SplitT(t[h][I.value]);
                });
            const auto *L1m = kc(d->lift_L1m);
            const auto *L2m = kc(d->lift_L2m);
            const auto row_out = [&](int j) {
                const u64 bp = d->b_p[j];
                const auto *row = L1m + j * k; // constants carry the factor 2^64: REDC removes it
                u64 v[NC];
#pragma unroll
                for (int h = 0; h < NC; h++)
                {
                    u64 temp = r_mt[h];
                    if (temp >= (1ull << 31))
                        temp += bp - (1ull << 32); // centred reduction of r_m_tilde, rns.cpp:969-973
                    u64 lo, hi;
                    if constexpr (KMAX < 0)
                    {
                        DotAcc<KA + 1> acc2; // carry-free accumulation (devmath.hpp), same integer sum
                        acc2.template add<0>(SplitT(temp), L2m[j]);
                        static_for<KA>([&](auto I) { acc2.template add<I.value + 1>(ts[h][I.value], row[I.value]); });
                        acc2.finish(lo, hi);
                    }
                    else
                    {
                        lo = temp * L2m[j];
                        hi = mulhi(temp, L2m[j]);
#pragma unroll
                        for (int i = 0; i < KA; i++)
                            if (i < k)
                                mac128(lo, hi, t[h][i], row[i]);
                    }
                    v[h] = redc_finish(redc128(lo, hi, bp, d->b_ninv[j]), bp, d->b_rdp[j], small);
                }
                if constexpr (TOP)
                {
                    butterfly_fwd_hs<true>(v[0], v[1], d->b_w1[j], d->b_w1s[j], 0 - bp, bp << 1);
                    pout[j * N + (N >> 1)] = v[1];
                }
                pout[j * N] = v[0];
            };
            if constexpr (KMAX < 0)
            {
                // |Bsk| is k + 1 or k + 2 (rns.cpp:568-573): with an exact K only the last row is a run-time question, the
                // others are straight-line code whose scalar loads overlap the neighbouring rows' arithmetic
                static_for<KA + 1>([&](auto J) { row_out(J.value); });
                if (nB == KA + 2)
                    row_out(KA + 1);
            }
            else
                for (int j = 0; j < nB; j++)
                    row_out(j);
        }

        // Constant-folded variant of bfv_floor_sk_kernel (k <= 32); see RnsDev for the folded constants.
        // Every output is the same canonical residue as the step-by-step version: only exact modular
        // identities are used ((b - c)*g == -c*g, (x*t)*g == x*(t*g), mul_add_mod(a,b,c) == a*b + c (mod q)).
        // One coefficient of an inverse-NTT output whose top layer was deferred (kNttDeferTop): apply
        // BackwardLazyLast (ntt.cpp:274-281) to the pair (c mod N/2, c mod N/2 + N/2) and the canonicalising
        // subtraction of ntt.h:328-333, keeping only this lane's side of the butterfly.
        template <class PP>
        __device__ __forceinline__ u64 after_top(u64 u, u64 v, bool is_hi, PP P)
        {
            const u64 p = P->p, two_p = P->two_p, neg_p = 0 - p;
            u64 r;
            if (is_hi)
                r = mulmod_lazy_hs<true>(u - v + two_p, P->inv_n_w, P->inv_n_w_shoup, neg_p);
            else
            {
                u64 tt = u + v;
                tt = tt >= two_p ? tt - two_p : tt;
                r = mulmod_lazy_hs<true>(tt, P->inv_n, P->inv_n_shoup, neg_p);
            }
            return r >= p ? r - p : r;
        }
        // The same pair without the multiplication: its constant (n^{-1} or w * n^{-1}) is folded into the constant the
        // value is multiplied with next (RnsDev::floor_F0_top / floor_G1m_top). REDUCE brings the value below 2p (operands
        // of the carry-free dot products must stay below 2^61); a Shoup product takes it as it is.
        template <bool REDUCE>
        __device__ __forceinline__ u64 before_top(u64 u, u64 v, bool is_hi, u64 two_p)
        {
            const u64 vv = is_hi ? two_p - v : v; // (a select, not a branch: a uniform branch here splits the row loop into
            u64 r = u + vv;                       //  blocks the scalar loads cannot be scheduled across); r < 4p
            if (REDUCE)
                r = r >= two_p ? r - two_p : r;
            return r;
        }
        template <int KMAX, bool DEFER>
        __global__ __launch_bounds__(kThreads) void bfv_floor_sk2_kernel(const RnsDev *__restrict__ d_,
                                                                         const PrimeDev *__restrict__ primes_,
                                                                         const u64 *__restrict__ in,
                                                                         std::size_t in_stride, u64 *__restrict__ out,
                                                                         std::size_t out_stride, std::size_t count,
                                                                         int logn, int mont, unsigned *__restrict__ tflags)
        {
            constexpr int KA = KMAX < 0 ? -KMAX : KMAX;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const std::size_t half = N >> 1;
            Cols cc;
            bool is_hi = false;
            if constexpr (DEFER)
            {
                // With the deferred top layer the lanes of columns c and c + N/2 read the same 2(k + |Bsk| + 1) words. Blocks
                // are dealt round-robin over the 8 XCDs (one L2 each): physical blocks b and b + 8 run on the same XCD at
                // about the same time, so they are given the lower and the upper half of the same 256 columns and the second
                // read comes from L2 instead of HBM (the halves used to be half a grid apart: 37 row passes instead of 22).
                // logn >= 14 here: the blocks of an item are a multiple of 16. Everything is derived from the block index, so
                // the compiler knows is_hi is uniform and reads the tables selected by it with scalar loads.
                const std::size_t per_item = N / kThreads, bid = blockIdx.x;
                cc.item = bid / per_item;
                if (cc.item >= count)
                    return;
                const unsigned pl = static_cast<unsigned>(bid - cc.item * per_item), r = pl & 15u;
                is_hi = (r >> 3) != 0;
                cc.c = (is_hi ? half : 0) + static_cast<std::size_t>(((pl >> 4) << 3) | (r & 7u)) * kThreads + threadIdx.x;
            }
            else if (!column(count, logn, cc))
                return;
            const auto *d = kc(d_);           // read-only for the lifetime of the context: constant address space
            const auto *primes = kc(primes_); // (scalar loads the compiler may merge and hoist)
            const int k = KMAX < 0 ? KA : d->k, B = d->B; // B is k or k + 1
            // with an exact K every test against B is a compile-time fact except for the one optional extra prime of B
            // (index K): no branches inside the row loops, so scalar loads and arithmetic of neighbouring rows overlap
            const bool extraB = B > KA;
            auto lt_B = [&](int j) { return KMAX < 0 ? (j < KA || (j == KA && extraB)) : j < B; };
            auto le_B = [&](int j) { return KMAX < 0 ? (j <= KA || (j == KA + 1 && extraB)) : j <= B; };
            const u64 *pin = in + cc.item * in_stride + cc.c;
            const u64 *pitem = in + cc.item * in_stride;
            const std::size_t c_lo = cc.c & (half - 1);
            u64 *pout = out + cc.item * out_stride + cc.c;
            // every input word of this column is requested before any arithmetic: one exposed memory latency per
            // thread instead of one per output row (the loads used to sit in the row loops)
            u64 ru[2 * KA + 2], rv[2 * KA + 2];
#pragma unroll
            for (int i = 0; i < KA; i++)
                if (KMAX < 0 || i < k)
                {
                    ru[i] = DEFER ? pitem[i * N + c_lo] : pin[i * N];
                    rv[i] = DEFER ? pitem[i * N + c_lo + half] : 0;
                }
            const auto load_bsk = [&](int j) {
                if (le_B(j))
                {
                    ru[KA + j] = DEFER ? pitem[(k + j) * N + c_lo] : pin[(k + j) * N];
                    rv[KA + j] = DEFER ? pitem[(k + j) * N + c_lo + half] : 0;
                }
            };
            if constexpr (KMAX < 0)
                static_for<KA + 2>([&](auto J) { load_bsk(J.value); });
            else
            {
#pragma unroll
                for (int j = 0; j < KA + 2; j++)
                    load_bsk(j);
            }
            __builtin_amdgcn_sched_barrier(0);
            u64 t[KA];
            // mont (kernel-uniform): the input words carry the Montgomery factor 2^-64 of the tensor product formed inside
            // the inverse NTT; the constants of the first product of every input then carry 2^64 on top
            const int top_sel = (mont ? 2 : 0) + (is_hi ? 1 : 0); // scalar: selects a table by address arithmetic
            const auto *F0t = DEFER ? kc(d->floor_F0_top[top_sel]) : kc(d->floor_F0);
            const auto *F0ts = DEFER ? kc(d->floor_F0_top_s[top_sel]) : kc(d->floor_F0_s);
            (void)primes;
            // the constants of the first loop, requested as one batch of scalar loads while the vector loads are in flight
            // (inside the loop every iteration waited for its own three)
            u64 cF0[KA], cF0s[KA], cQ[KA];
#pragma unroll
            for (int i = 0; i < KA; i++)
                if (KMAX < 0 || i < k)
                {
                    cF0[i] = F0t[i];
                    cF0s[i] = F0ts[i];
                    cQ[i] = d->q_p[i];
                }
            __builtin_amdgcn_sched_barrier(0);
#pragma unroll
            for (int i = 0; i < KA; i++)
                if (KMAX < 0 || i < k)
                {
                    const u64 qp = cQ[i];
                    if (DEFER) // (u +- v) * (n^{-1} or w n^{-1}) * F0 with ONE canonical Shoup product
                        t[i] = mulmod_shoup_hs(before_top<false>(ru[i], rv[i], is_hi, qp << 1), cF0[i], cF0s[i], qp);
                    else
                        t[i] = mulmod_shoup_hs(ru[i], cF0[i], cF0s[i], qp);
                }
            u64 tb[KA + 1];
            u64 fl_sk = 0;
            // exact-K instances are only launched when the host proved every REDC lands below 2p (RnsDev::redc_small):
            // a compile-time fact there, so the row loops carry no branch
            const bool small = KMAX < 0 ? true : d->redc_small != 0;
            SplitT ts[KA], tbs[KA + 1];
            if constexpr (KMAX < 0)
                static_for<KA>([&](auto I) { ts[I.value] = SplitT(t[I.value]); });
            const auto *G1m = DEFER ? kc(d->floor_G1m_top[top_sel]) : kc(d->floor_G1m);
            const auto *G2m = kc(d->floor_G2m);
            const auto bsk_row = [&](int j) {
                if (le_B(j))
                {
                    const u64 bp = d->b_p[j];
                    const auto *row = G2m + j * k;
                    const u64 x = DEFER ? before_top<true>(ru[KA + j], rv[KA + j], is_hi, bp << 1) : ru[KA + j];
                    u64 lo, hi;
                    if constexpr (KMAX < 0)
                    {
                        DotAcc<KA + 1> acc2;
                        acc2.template add<0>(SplitT(x), G1m[j]);
                        static_for<KA>([&](auto I) { acc2.template add<I.value + 1>(ts[I.value], row[I.value]); });
                        acc2.finish(lo, hi);
                    }
                    else
                    {
                        lo = x * G1m[j];
                        hi = mulhi(x, G1m[j]);
#pragma unroll
                        for (int i = 0; i < KA; i++)
                            if (i < k)
                                mac128(lo, hi, t[i], row[i]);
                    }
                    const u64 v = redc_finish(redc128(lo, hi, bp, d->b_ninv[j]), bp, d->b_rdp[j], small);
                    if (lt_B(j))
                        tb[j < KA + 1 ? j : 0] = v;
                    else
                        fl_sk = v;
                    // one row's constants at a time: left alone the scheduler requests the constants of every row up front
                    // and spills scalar registers to scratch
                    __builtin_amdgcn_sched_barrier(0);
                }
            };
            // (compile-time row indices for an exact K: a `#pragma unroll` gives up on these loops from K = 15 on and leaves
            //  run-time indexed register arrays, i.e. scratch)
            if constexpr (KMAX < 0)
                static_for<KA + 2>([&](auto J) { bsk_row(J.value); });
            else
                for (int j = 0; j < KA + 2; j++)
                    bsk_row(j);
            const u64 mp = d->b_p[B];
            const auto *BtoMsk = kc(d->B_to_mskm);
            u64 lo = 0, hi = 0;
            if constexpr (KMAX < 0)
            {
                static_for<KA + 1>([&](auto J) {
                    if (lt_B(J.value))
                        tbs[J.value] = SplitT(tb[J.value]);
                });
                DotAcc<KA + 1> acc2;
                static_for<KA + 1>([&](auto J) {
                    if (lt_B(J.value))
                        acc2.template add<J.value>(tbs[J.value], BtoMsk[J.value]);
                });
                acc2.finish(lo, hi);
            }
            else
            {
#pragma unroll
                for (int j = 0; j < KA + 1; j++)
                    if (lt_B(j))
                        mac128(lo, hi, tb[j], BtoMsk[j]);
            }
            const u64 conv_sk = redc_finish(redc128(lo, hi, mp, d->b_ninv[B]), mp, d->b_rdp[B], small);
            const u64 alpha = mulmod_shoup(conv_sk + (mp - fl_sk), d->inv_prod_B_mod_msk, d->inv_prod_B_mod_msk_s, mp);
            const bool neg = alpha > (mp >> 1); // rns.cpp:909
            const u64 a2 = neg ? mp - alpha : alpha;
            const SplitT a2s(a2);
            const auto *pBm = kc(d->pBm);
            const auto *nBm = kc(d->nBm);
            const auto *BtoQ = kc(d->B_to_qm);
            u64 nz = 0; // OR of the words this column stores (transparency sink)
            const auto q_row = [&](int i) {
                const u64 c = neg ? pBm[i] : nBm[i];
                const auto *mrow = BtoQ + i * B;
                u64 l2, h2;
                if constexpr (KMAX < 0)
                {
                    DotAcc<KA + 2> acc2;
                    acc2.template add<0>(a2s, c);
                    static_for<KA + 1>([&](auto J) {
                        if (lt_B(J.value))
                            acc2.template add<J.value + 1>(tbs[J.value], mrow[J.value]);
                    });
                    acc2.finish(l2, h2);
                }
                else
                {
                    l2 = a2 * c;
                    h2 = mulhi(a2, c);
#pragma unroll
                    for (int j = 0; j < KA + 1; j++)
                        if (lt_B(j))
                            mac128(l2, h2, tb[j], mrow[j]);
                }
                const u64 qp = d->q_p[i];
                const u64 w = redc_finish(redc128(l2, h2, qp, d->q_ninv[i]), qp, d->q_rdp[i], small);
                pout[i * N] = w;
                nz |= w;
                __builtin_amdgcn_sched_barrier(0);
            };
            if constexpr (KMAX < 0)
                static_for<KA>([&](auto I) { q_row(I.value); });
            else
                for (int i = 0; i < k; i++)
                    q_row(i);
            note_nonzero(tflags, cc.item, nz); // (the launcher passes the sink for polynomials 1.. of the product only)
        }

        // fast_floor (rns.cpp:983-1023), optionally preceded by the multiplication by t of
        // evaluator.cpp:432-434: (k + |Bsk|) rows -> |Bsk| rows
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void fast_floor_kernel(const RnsDev *__restrict__ d,
                                                                      const PrimeDev *__restrict__ primes,
                                                                      const u64 *__restrict__ in, std::size_t in_stride,
                                                                      u64 *__restrict__ out, std::size_t out_stride,
                                                                      std::size_t count, int logn, int mul_t)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k, nB = d->nB;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            u64 t[KMAX];
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                {
                    const PrimeDev &Q = primes[d->q_prime[i]];
                    u64 x = pin[i * N];
                    if (mul_t)
                        x = mul_mod(x, d->t, Q.p, Q.cr0, Q.cr1);
                    t[i] = mul_mod(x, d->q_inv[i], Q.p, Q.cr0, Q.cr1);
                }
            for (int j = 0; j < nB; j++)
            {
                const PrimeDev &Bp = primes[d->bsk_prime[j]];
                const u64 conv = dot_mod<KMAX>(t, k, d->q_to_Bsk + j * k, Bp);
                u64 x = pin[(k + j) * N];
                if (mul_t)
                    x = mul_mod(x, d->t, Bp.p, Bp.cr0, Bp.cr1);
                pout[j * N] = mul_mod(x + (Bp.p - conv), d->inv_prod_q_mod_Bsk[j], Bp.p, Bp.cr0, Bp.cr1);
            }
        }

        // Shenoy-Kumaresan tail shared by fastbconv_sk and the fused kernel (rns.cpp:880-922)
        template <int KMAX>
        __device__ __forceinline__ void sk_finish(const RnsDev *d, const PrimeDev *primes, const u64 (&tb)[KMAX + 1],
                                                  u64 in_sk, u64 *pout, std::size_t N)
        {
            const int k = d->k, B = d->B;
            const PrimeDev &Msk = primes[d->bsk_prime[B]];
            u64 lo = 0, hi = 0;
#pragma unroll
            for (int i = 0; i < KMAX + 1; i++)
                if (i < B)
                    mac128(lo, hi, tb[i], d->B_to_msk[i]);
            const u64 conv_sk = barrett_reduce_128(lo, hi, Msk.p, Msk.cr0, Msk.cr1);
            const u64 alpha = mul_mod(conv_sk + (Msk.p - in_sk), d->inv_prod_B_mod_msk, Msk.p, Msk.cr0, Msk.cr1);
            const bool neg = alpha > (Msk.p >> 1);
            for (int i = 0; i < k; i++)
            {
                const PrimeDev &Q = primes[d->q_prime[i]];
                u64 l2 = 0, h2 = 0;
                const u64 *mrow = d->B_to_q + i * B;
#pragma unroll
                for (int a = 0; a < KMAX + 1; a++)
                    if (a < B)
                        mac128(l2, h2, tb[a], mrow[a]);
                const u64 conv = barrett_reduce_128(l2, h2, Q.p, Q.cr0, Q.cr1);
                const u64 pB = d->prod_B_mod_q[i];
                pout[i * N] = neg ? mul_add_mod(pB, Msk.p - alpha, conv, Q.p, Q.cr0, Q.cr1)
                                  : mul_add_mod(Q.p - pB, alpha, conv, Q.p, Q.cr0, Q.cr1);
            }
        }

        // fastbconv_sk (rns.cpp:853-923): |Bsk| rows -> k rows
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void fastbconv_sk_kernel(const RnsDev *__restrict__ d,
                                                                        const PrimeDev *__restrict__ primes,
                                                                        const u64 *__restrict__ in,
                                                                        std::size_t in_stride, u64 *__restrict__ out,
                                                                        std::size_t out_stride, std::size_t count,
                                                                        int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int B = d->B;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 tb[KMAX + 1];
#pragma unroll
            for (int i = 0; i < KMAX + 1; i++)
                if (i < B)
                {
                    const PrimeDev &Bp = primes[d->bsk_prime[i]];
                    tb[i] = mul_mod(pin[i * N], d->B_inv[i], Bp.p, Bp.cr0, Bp.cr1);
                }
            sk_finish<KMAX>(d, primes, tb, pin[B * N], out + cc.item * out_stride + cc.c, N);
        }

        // fused BEHZ steps (6)-(8) of evaluator.cpp:427-444: (k+|Bsk|) rows -> k rows
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void bfv_floor_sk_kernel(const RnsDev *__restrict__ d,
                                                                        const PrimeDev *__restrict__ primes,
                                                                        const u64 *__restrict__ in,
                                                                        std::size_t in_stride, u64 *__restrict__ out,
                                                                        std::size_t out_stride, std::size_t count,
                                                                        int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k, B = d->B;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 t[KMAX];
#pragma unroll
            for (int i = 0; i < KMAX; i++)
                if (i < k)
                {
                    const PrimeDev &Q = primes[d->q_prime[i]];
                    const u64 x = mul_mod(pin[i * N], d->t, Q.p, Q.cr0, Q.cr1);
                    t[i] = mul_mod(x, d->q_inv[i], Q.p, Q.cr0, Q.cr1);
                }
            u64 tb[KMAX + 1];
            u64 in_sk = 0;
#pragma unroll
            for (int j = 0; j < KMAX + 2; j++)
                if (j <= B)
                {
                    const PrimeDev &Bp = primes[d->bsk_prime[j]];
                    const u64 conv = dot_mod<KMAX>(t, k, d->q_to_Bsk + j * k, Bp);
                    const u64 x = mul_mod(pin[(k + j) * N], d->t, Bp.p, Bp.cr0, Bp.cr1);
                    const u64 fl = mul_mod(x + (Bp.p - conv), d->inv_prod_q_mod_Bsk[j], Bp.p, Bp.cr0, Bp.cr1);
                    if (j < B)
                    {
                        if (j < KMAX + 1)
                            tb[j < KMAX + 1 ? j : 0] = mul_mod(fl, d->B_inv[j], Bp.p, Bp.cr0, Bp.cr1);
                    }
                    else
                        in_sk = fl;
                }
            sk_finish<KMAX>(d, primes, tb, in_sk, out + cc.item * out_stride + cc.c, N);
        }

        // divide_and_round_q_last_inplace (rns.cpp:731-775); writes out_rows rows (k-1, or k to mirror the
        // reference's clobbered last row)
        __global__ __launch_bounds__(kThreads) void divround_bfv_kernel(const RnsDev *__restrict__ d,
                                                                        const PrimeDev *__restrict__ primes,
                                                                        const u64 *__restrict__ in,
                                                                        std::size_t in_stride, u64 *__restrict__ out,
                                                                        std::size_t out_stride, std::size_t count,
                                                                        int logn, int out_rows)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            const PrimeDev &L = primes[d->q_prime[k - 1]];
            const u64 half = L.p >> 1;
            const u64 last = barrett_reduce_63(pin[(k - 1) * N] + half, L.p, L.cr1);
            for (int i = 0; i < k - 1; i++)
            {
                const PrimeDev &Q = primes[d->q_prime[i]];
                u64 temp = barrett_reduce_63(last, Q.p, Q.cr1);
                temp = sub_mod(temp, barrett_reduce_63(half, Q.p, Q.cr1), Q.p);
                const u64 v = sub_mod(pin[i * N], temp, Q.p);
                pout[i * N] = mul_mod(v, d->inv_q_last_mod_q[i], Q.p, Q.cr0, Q.cr1);
            }
            if (out_rows == k)
                pout[(k - 1) * N] = last;
        }

        // divide_and_round_q_last_ntt_inplace, part before the forward NTTs (rns.cpp:800-827); `last` is the
        // already inverse-transformed last row and is updated in place like the reference does
        __global__ __launch_bounds__(kThreads) void rescale_pre_kernel(const RnsDev *__restrict__ d,
                                                                       const PrimeDev *__restrict__ primes,
                                                                       u64 *__restrict__ last_row,
                                                                       std::size_t last_stride,
                                                                       u64 *__restrict__ temp, std::size_t temp_stride,
                                                                       std::size_t count, int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            u64 *plast = last_row + cc.item * last_stride + cc.c;
            u64 *ptemp = temp + cc.item * temp_stride + cc.c;
            const PrimeDev &L = primes[d->q_prime[k - 1]];
            const u64 half = L.p >> 1;
            const u64 last = barrett_reduce_63(*plast + half, L.p, L.cr1);
            *plast = last;
            for (int i = 0; i < k - 1; i++)
            {
                const PrimeDev &Q = primes[d->q_prime[i]];
                const u64 v = Q.p < L.p ? barrett_reduce_63(last, Q.p, Q.cr1) : last;
                ptemp[i * N] = v + (Q.p - barrett_reduce_63(half, Q.p, Q.cr1));
            }
        }

        // ... and the part after them (rns.cpp:841-849)
        __global__ __launch_bounds__(kThreads) void rescale_post_kernel(const RnsDev *__restrict__ d,
                                                                        const PrimeDev *__restrict__ primes,
                                                                        const u64 *__restrict__ in,
                                                                        std::size_t in_stride,
                                                                        const u64 *__restrict__ temp,
                                                                        std::size_t temp_stride, u64 *__restrict__ out,
                                                                        std::size_t out_stride, std::size_t count,
                                                                        int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * in_stride + cc.c;
            const u64 *ptemp = temp + cc.item * temp_stride + cc.c;
            u64 *pout = out + cc.item * out_stride + cc.c;
            for (int i = 0; i < k - 1; i++)
            {
                const PrimeDev &Q = primes[d->q_prime[i]];
                const u64 v = pin[i * N] + (Q.p << 2) - ptemp[i * N];
                pout[i * N] = mul_mod(v, d->inv_q_last_mod_q[i], Q.p, Q.cr0, Q.cr1);
            }
        }

        inline unsigned blocks_for(std::size_t count, int logn)
        {
            return static_cast<unsigned>(((count << logn) + kThreads - 1) / kThreads);
        }

        // RNSTool::decrypt_scale_and_round (rns.cpp:1070-1126): k rows -> one row of coefficients mod t. The two scalar
        // multiplications before the base conversion are folded into one constant; everything else as written there.
        __global__ __launch_bounds__(kThreads) void decrypt_scale_and_round_kernel(const RnsDev *__restrict__ d,
                                                                                   const PrimeDev *__restrict__ primes,
                                                                                   const u64 *__restrict__ in,
                                                                                   u64 *__restrict__ out,
                                                                                   std::size_t count, int logn)
        {
            Cols cc;
            if (!column(count, logn, cc))
                return;
            const int k = d->k;
            const std::size_t N = static_cast<std::size_t>(1) << logn;
            const u64 *pin = in + cc.item * (static_cast<std::size_t>(k) * N) + cc.c;
            const u64 t = d->t, gamma = d->dsr_gamma;
            const PrimeDev &G = primes[d->gamma_prime];
            u64 lt = 0, ht = 0, lg = 0, hg = 0;
            for (int i = 0; i < k; i++)
            {
                const u64 qi = primes[d->q_prime[i]].p;
                const u64 y = mulmod_shoup(pin[i * N], d->dsr_scale[i], d->dsr_scale_s[i], qi); // :1076-1082 + :476-485
                mac128(lt, ht, y, d->dsr_to_t[i]);                                             // :487-495
                mac128(lg, hg, y, d->dsr_to_g[i]);
            }
            // reductions mod t (any modulus below 2^61: plain 128-bit remainder) and mod gamma (Barrett)
            const unsigned __int128 st = (static_cast<unsigned __int128>(ht) << 64) | lt;
            u64 vt = static_cast<u64>(st % t);
            u64 vg = barrett_reduce_128(lg, hg, G.p, G.cr0, G.cr1);
            vt = static_cast<u64>((static_cast<unsigned __int128>(vt) * d->dsr_neg_inv_q_t) % t); // :1091-1096
            vg = mul_mod(vg, d->dsr_neg_inv_q_g, G.p, G.cr0, G.cr1);
            u64 r;
            if (vg > (gamma >> 1)) // :1107-1117
            {
                const u64 a = vt + (gamma - vg) % t;
                r = a >= t ? a - t : a;
            }
            else
            {
                const u64 b = vg % t;
                r = vt >= b ? vt - b : vt + t - b;
            }
            if (r) // :1120-1124
                r = static_cast<u64>((static_cast<unsigned __int128>(r) * d->dsr_inv_gamma_t) % t);
            out[cc.item * N + cc.c] = r;
        }
    } // namespace

#define SEALHIP_DISPATCH_K(kval, KERNEL, ...)                                             \
    do                                                                                    \
    {                                                                                     \
        if ((kval) <= 4)                                                                  \
            KERNEL<4><<<grid, kThreads, 0, e.lane().stream>>>(__VA_ARGS__);                      \
        else if ((kval) <= 8)                                                             \
            KERNEL<8><<<grid, kThreads, 0, e.lane().stream>>>(__VA_ARGS__);                      \
        else if ((kval) <= 16)                                                            \
            KERNEL<16><<<grid, kThreads, 0, e.lane().stream>>>(__VA_ARGS__);                     \
        else if ((kval) <= 32)                                                            \
            KERNEL<32><<<grid, kThreads, 0, e.lane().stream>>>(__VA_ARGS__);                     \
        else                                                                              \
            KERNEL<64><<<grid, kThreads, 0, e.lane().stream>>>(__VA_ARGS__);                     \
    } while (0)

    hipError_t launch_fastbconv_m_tilde(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                        std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count)
    {
        if (!count)
            return hipSuccess;
        const unsigned grid = blocks_for(count, e.logn);
        ProfScope prof(e, "fastbconv_m_tilde", 0);
        SEALHIP_DISPATCH_K(h.k, fastbconv_m_tilde_kernel, d, e.d_primes, in, in_stride, out, out_stride, count, e.logn);
        return hipGetLastError();
    }

    hipError_t launch_sm_mrq(const Engine &e, const RnsDev *d, const RnsDev &, const u64 *in, std::size_t in_stride,
                             u64 *out, std::size_t out_stride, std::size_t count)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "sm_mrq", 0);
        sm_mrq_kernel<<<blocks_for(count, e.logn), kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out,
                                                                            out_stride, count, e.logn);
        return hipGetLastError();
    }

    namespace
    {
        template <int KM>
        void lift2_launch(const Engine &e, const RnsDev *d, const u64 *in, std::size_t in_stride, u64 *out, std::size_t out_stride,
                          std::size_t count, unsigned grid, bool top_layer)
        {
            if constexpr (KM < 0)
            {
                if (top_layer)
                {
                    bfv_lift2_kernel<KM, true><<<grid, kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out, out_stride,
                                                                                       count, e.logn);
                    return;
                }
            }
            bfv_lift2_kernel<KM><<<grid, kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out, out_stride, count, e.logn);
        }
    } // namespace

    bool bfv_lift_can_apply_top(const Engine &e, const RnsDev &h)
    {
        static const bool off = exp_env("SEALHIP_LIFT_TOP_OFF") != nullptr; // (measurement-only build)
        // (STRICT: the caller also asks ntt_strict_top_done_ok -- only the dense forward schedule starts below the top layer;
        //  the layer itself is the same butterfly on canonical words either way: nothing can wrap in it)
        return !off && h.redc_small && h.k >= 1 && h.k <= 16 && e.logn >= 14;
    }

    hipError_t launch_bfv_lift(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in, std::size_t in_stride,
                               u64 *out, std::size_t out_stride, std::size_t count, bool top_layer)
    {
        if (!count)
            return hipSuccess;
        if (top_layer && !bfv_lift_can_apply_top(e, h))
            return hipErrorInvalidValue;
        const unsigned grid = blocks_for(count, top_layer ? e.logn - 1 : e.logn);
        ProfScope prof(e, "bfv_lift", 0);
        if (h.k <= 32)
        {
#define SEALHIP_LIFT2(KM) lift2_launch<KM>(e, d, in, in_stride, out, out_stride, count, grid, top_layer)
            switch (h.redc_small ? h.k : 0)
            {
            case 1: SEALHIP_LIFT2(-1); break;
            case 2: SEALHIP_LIFT2(-2); break;
            case 3: SEALHIP_LIFT2(-3); break;
            case 4: SEALHIP_LIFT2(-4); break;
            case 5: SEALHIP_LIFT2(-5); break;
            case 6: SEALHIP_LIFT2(-6); break;
            case 7: SEALHIP_LIFT2(-7); break;
            case 8: SEALHIP_LIFT2(-8); break;
            case 9: SEALHIP_LIFT2(-9); break;
            case 10: SEALHIP_LIFT2(-10); break;
            case 11: SEALHIP_LIFT2(-11); break;
            case 12: SEALHIP_LIFT2(-12); break;
            case 13: SEALHIP_LIFT2(-13); break;
            case 14: SEALHIP_LIFT2(-14); break;
            case 15: SEALHIP_LIFT2(-15); break;
            case 16: SEALHIP_LIFT2(-16); break;
            default: SEALHIP_LIFT2(32); break;
            }
#undef SEALHIP_LIFT2
            return hipGetLastError();
        }
        SEALHIP_DISPATCH_K(h.k, bfv_lift_kernel, d, e.d_primes, in, in_stride, out, out_stride, count, e.logn);
        return hipGetLastError();
    }

    hipError_t launch_fast_floor(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                 std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count, int mul_t)
    {
        if (!count)
            return hipSuccess;
        const unsigned grid = blocks_for(count, e.logn);
        ProfScope prof(e, "fast_floor", 0);
        SEALHIP_DISPATCH_K(h.k, fast_floor_kernel, d, e.d_primes, in, in_stride, out, out_stride, count, e.logn,
                           mul_t);
        return hipGetLastError();
    }

    hipError_t launch_fastbconv_sk(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count)
    {
        if (!count)
            return hipSuccess;
        const unsigned grid = blocks_for(count, e.logn);
        ProfScope prof(e, "fastbconv_sk", 0);
        SEALHIP_DISPATCH_K(h.k, fastbconv_sk_kernel, d, e.d_primes, in, in_stride, out, out_stride, count, e.logn);
        return hipGetLastError();
    }

    hipError_t launch_bfv_floor_sk(const Engine &e, const RnsDev *d, const RnsDev &h, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                   int deferred_top)
    {
        if (!count)
            return hipSuccess;
        const unsigned grid = blocks_for(count, e.logn);
        // transparency sink armed for this launch (pipeline.cpp SinkArm: polynomials 1.. of a product): the fused kernel
        // notes non-zero words as it stores them; the step-by-step kernels are followed by the read pass instead
        unsigned *tflags = e.lane().tsink_arm;
        ProfScope prof(e, "bfv_floor_sk", 0);
        if (h.k <= 32)
        {
#define SEALHIP_FLOOR2(KM)                                                                                          \
    do                                                                                                               \
    {                                                                                                                \
        if (deferred_top)                                                                                            \
            bfv_floor_sk2_kernel<KM, true><<<grid, kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out,       \
                                                                           out_stride, count, e.logn,               \
                                                                           deferred_top == 2 ? 1 : 0, tflags);      \
        else                                                                                                         \
            bfv_floor_sk2_kernel<KM, false><<<grid, kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out,      \
                                                                            out_stride, count, e.logn, 0, tflags);  \
    } while (0)
            switch (h.redc_small ? h.k : 0)
            {
            case 1: SEALHIP_FLOOR2(-1); break;
            case 2: SEALHIP_FLOOR2(-2); break;
            case 3: SEALHIP_FLOOR2(-3); break;
            case 4: SEALHIP_FLOOR2(-4); break;
            case 5: SEALHIP_FLOOR2(-5); break;
            case 6: SEALHIP_FLOOR2(-6); break;
            case 7: SEALHIP_FLOOR2(-7); break;
            case 8: SEALHIP_FLOOR2(-8); break;
            case 9: SEALHIP_FLOOR2(-9); break;
            case 10: SEALHIP_FLOOR2(-10); break;
            case 11: SEALHIP_FLOOR2(-11); break;
            case 12: SEALHIP_FLOOR2(-12); break;
            case 13: SEALHIP_FLOOR2(-13); break;
            case 14: SEALHIP_FLOOR2(-14); break;
            case 15: SEALHIP_FLOOR2(-15); break;
            case 16: SEALHIP_FLOOR2(-16); break;
            default: SEALHIP_FLOOR2(32); break;
            }
#undef SEALHIP_FLOOR2
            return hipGetLastError();
        }
        if (deferred_top)
            return hipErrorInvalidValue;
        SEALHIP_DISPATCH_K(h.k, bfv_floor_sk_kernel, d, e.d_primes, in, in_stride, out, out_stride, count, e.logn);
        hipError_t err = hipGetLastError();
        if (err == hipSuccess && tflags)
            err = launch_nonzero_words(e, out, out_stride, static_cast<std::size_t>(h.k) << e.logn, count, tflags);
        return err;
    }

    hipError_t launch_divround_bfv(const Engine &e, const RnsDev *d, const RnsDev &, const u64 *in,
                                   std::size_t in_stride, u64 *out, std::size_t out_stride, std::size_t count,
                                   int out_rows)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "divround_bfv", 0);
        divround_bfv_kernel<<<blocks_for(count, e.logn), kThreads, 0, e.lane().stream>>>(d, e.d_primes, in, in_stride, out,
                                                                                  out_stride, count, e.logn, out_rows);
        return hipGetLastError();
    }

    hipError_t launch_rescale_pre(const Engine &e, const RnsDev *d, const RnsDev &, u64 *last,
                                  std::size_t last_stride, u64 *temp, std::size_t temp_stride, std::size_t count)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "rescale_pre", 0);
        rescale_pre_kernel<<<blocks_for(count, e.logn), kThreads, 0, e.lane().stream>>>(d, e.d_primes, last, last_stride,
                                                                                 temp, temp_stride, count, e.logn);
        return hipGetLastError();
    }

    hipError_t launch_rescale_post(const Engine &e, const RnsDev *d, const RnsDev &, const u64 *in,
                                   std::size_t in_stride, const u64 *temp, std::size_t temp_stride, u64 *out,
                                   std::size_t out_stride, std::size_t count)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "rescale_post", 0);
        rescale_post_kernel<<<blocks_for(count, e.logn), kThreads, 0, e.lane().stream>>>(
            d, e.d_primes, in, in_stride, temp, temp_stride, out, out_stride, count, e.logn);
        return hipGetLastError();
    }
    hipError_t launch_decrypt_scale_and_round(const Engine &e, const RnsDev *d, const RnsDev &, const u64 *in, u64 *out,
                                              std::size_t count)
    {
        if (!count)
            return hipSuccess;
        ProfScope prof(e, "decrypt_scale_and_round", 0);
        const std::size_t lanes = count << e.logn;
        decrypt_scale_and_round_kernel<<<static_cast<unsigned>((lanes + kThreads - 1) / kThreads), kThreads, 0, e.lane().stream>>>(
            d, e.d_primes, in, out, count, e.logn);
        return hipGetLastError();
    }
} // namespace sealhip
