// ntt.hip -- batched negacyclic NTT / inverse NTT for gfx950 (CDNA4).
//
// Replaces util::ntt_negacyclic_harvey{,_lazy} / inverse_ntt_negacyclic_harvey{,_lazy}
// (native/src/seal/util/ntt.cpp:292-404, ntt.h:225-334) for batches of RNS rows.
//
// Shape of the computation: integer, HBM-streaming, ALU-heavy (one Shoup butterfly = one
// 64x64->hi64 and two 64x64->lo64 products built from v_mad_u64_u32); no MFMA.
//
// Decomposition: a row of N = 2^logn coefficients is transformed in ONE launch: for logn <= 13 by the
// tiled kernel (ntt_pass_kernel: the row is a TILE of 2^t coefficients staged HBM -> LDS with 16-byte
// coalesced loads, the threads run register-resident radix-16 rounds -- 16 coefficients = 4 index bits per
// thread, up to 4 butterfly layers per LDS round trip -- and the tile is stored back with 16-byte coalesced
// stores), for logn 14..16 by the single-pass half-row kernels further down. (The two-launch "strided
// columns, then contiguous rows" split of round 1 for logn > 13 was removed in round 4 with its switch.)
// Bit-exactness: every butterfly is exactly the reference's radix-2 lazy butterfly,
// only the schedule differs, so the 64-bit words (including the wrap-around behaviour for 60-bit
// primes) are identical. Twiddle of the butterfly on global bit b whose lower element
// has index j: table entry (N + j) >> (b + 1), in both directions.
//
// LDS image: index l is stored at l + (l >> 4) (one pad word per 16), which makes both the
// stride-1 and the stride-16 access patterns of the rounds bank-conflict free for ds_read_b64.
#include <cstring>
#include <cstdlib>

#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        constexpr int kTileBitsMax = 13; // 2^13 coefficients = 64 KiB (+4 KiB pad): two workgroups per CU

        __device__ __forceinline__ int pad_index(int l)
        {
            return l + (l >> 4);
        }

        // DIR 0: forward (Cooley-Tukey, descending bits), DIR 1: inverse (Gentleman-Sande, ascending bits)
        template <int DIR>
        __global__ __launch_bounds__(512, 4) void ntt_pass_kernel(u64 *__restrict__ data,
                                                               const PrimeDev *__restrict__ primes, RowMap map,
                                                               NttPass ps)
        {
            extern __shared__ u64 lds[];
            const int tid = threadIdx.x;
            const int nthreads = blockDim.x;
            const int t = ps.t, c = ps.c, b_lo = ps.b_lo, logn = ps.logn;
            const int tile = blockIdx.x & ((1 << (logn - t)) - 1);
            const size_t row = blockIdx.x >> (logn - t);
            const unsigned short pid = map.prime[row % map.rows];
            if (pid == kSkipRow)
                return; // block-uniform: this row is not part of the transform (e.g. in-bundle rows)
            const PrimeDev P = primes[pid];
            const u64 p = P.p, two_p = P.two_p;
            u64 *rowp = data + (row << logn);
            const int cmask = (1 << c) - 1;
            const int mid_bits = b_lo - c;
            const int base = ((tile >> mid_bits) << (b_lo + t - c)) | ((tile & ((1 << mid_bits) - 1)) << c);
            const u64 *tw = DIR == 0 ? P.fwd : P.inv;
            const int N = 1 << logn;
            const bool strict = (ps.flags & kNttStrict) != 0;

            // ---- stage in: 8 x 16-byte loads per thread, coalesced along the contiguous part of the tile
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                int l = 2 * (i * nthreads + tid);
                int g = base + ((l >> c) << b_lo) + (l & cmask);
                const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rowp + g);
                int a = pad_index(l);
                lds[a] = v.x;
                lds[a + 1] = v.y;
            }
            __syncthreads();

            for (int r = 0; r < ps.nrounds; r++)
            {
                const int beta = ps.rounds[r].beta, wlo = ps.rounds[r].wlo, whi = ps.rounds[r].whi;
                const int low = tid & ((1 << beta) - 1);
                const int l0 = ((tid >> beta) << (beta + 4)) | low;
                u64 x[16];
#pragma unroll
                for (int s = 0; s < 16; s++)
                    x[s] = lds[pad_index(l0 + (s << beta))];
                const int j0 = base + ((l0 >> c) << b_lo) + (l0 & cmask);

#pragma unroll
                for (int step = 0; step < 4; step++)
                {
                    const int w = DIR == 0 ? 3 - step : step;
                    if (w < wlo || w > whi)
                        continue; // wave-uniform
                    const int lb = beta + w;
                    const int gb = lb < c ? lb : lb - c + b_lo; // global bit of this layer
                    const int tb = (N + j0) >> (gb + 1);
                    const int bit = 1 << w;
                    if (DIR == 0)
                    {
                        const bool last = (gb == 0) && !strict;
                        // eight butterflies per layer, four at a time in lock step (devmath.hpp: butterflies_fwd_hs)
#pragma unroll
                        for (int c4 = 0; c4 < 8; c4 += 4)
                        {
                            u64 uu[4], yy[4], ww[4], ws[4];
#pragma unroll
                            for (int j = 0; j < 4; j++)
                            {
                                const int s = (((c4 + j) >> w) << (w + 1)) | ((c4 + j) & (bit - 1)); // (c4+j)-th slot with bit w clear
                                const ulonglong2 W = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (tb + (s >> (w + 1))));
                                ww[j] = W.x;
                                ws[j] = W.y;
                                u64 u = x[s];
                                if (strict)
                                    u = u >= two_p ? u - two_p : u;
                                else if (last)
                                    u = barrett_lazy(u, P.rdp, p); // ForwardLazyLast, ntt.cpp:254-261
                                uu[j] = u;
                                yy[j] = x[s | bit];
                            }
                            butterflies_fwd_hs<false, 4>(uu, yy, ww, ws, 0 - p, two_p); // ForwardLazy, ntt.cpp:245-252
#pragma unroll
                            for (int j = 0; j < 4; j++)
                            {
                                const int s = (((c4 + j) >> w) << (w + 1)) | ((c4 + j) & (bit - 1));
                                x[s] = uu[j];
                                x[s | bit] = yy[j];
                            }
                        }
                    }
                    else
                    {
                        // Gentleman-Sande layer, four butterflies in lock step; the top layer (gap N/2) uses the merged
                        // twiddle psi^-1... * n^-1 on the difference side and multiplies the sum side by n^-1 afterwards
                        // (BackwardLazyLast, ntt.cpp:274-281)
                        const bool top = gb == logn - 1;
#pragma unroll
                        for (int c4 = 0; c4 < 8; c4 += 4)
                        {
                            u64 uu[4], yy[4], ww[4], ws[4];
#pragma unroll
                            for (int j = 0; j < 4; j++)
                            {
                                const int s = (((c4 + j) >> w) << (w + 1)) | ((c4 + j) & (bit - 1));
                                ulonglong2 W;
                                if (top)
                                {
                                    W.x = P.inv_n_w;
                                    W.y = P.inv_n_w_shoup;
                                }
                                else
                                    W = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (tb + (s >> (w + 1))));
                                ww[j] = W.x;
                                ws[j] = W.y;
                                uu[j] = x[s];
                                yy[j] = x[s | bit];
                            }
                            butterflies_inv_hs<false, 4>(uu, yy, ww, ws, 0 - p, two_p); // BackwardLazy, ntt.cpp:265-272
#pragma unroll
                            for (int j = 0; j < 4; j++)
                            {
                                const int s = (((c4 + j) >> w) << (w + 1)) | ((c4 + j) & (bit - 1));
                                x[s] = top ? mulmod_lazy(uu[j], P.inv_n, P.inv_n_shoup, p) : uu[j];
                                x[s | bit] = yy[j];
                            }
                        }
                    }
                }
#pragma unroll
                for (int s = 0; s < 16; s++)
                    lds[pad_index(l0 + (s << beta))] = x[s];
                __syncthreads();
            }

            // ---- stage out (optionally with the canonicalising wrapper of ntt.h:236-245 / :328-333)
            const bool canon = (ps.flags & kNttCanonical) != 0;
#pragma unroll
            for (int i = 0; i < 8; i++)
            {
                int l = 2 * (i * nthreads + tid);
                int g = base + ((l >> c) << b_lo) + (l & cmask);
                int a = pad_index(l);
                ulonglong2 v;
                v.x = lds[a];
                v.y = lds[a + 1];
                if (canon)
                {
                    if (DIR == 0)
                    {
                        v.x = v.x >= two_p ? v.x - two_p : v.x;
                        v.y = v.y >= two_p ? v.y - two_p : v.y;
                    }
                    v.x = v.x >= p ? v.x - p : v.x;
                    v.y = v.y >= p ? v.y - p : v.y;
                }
                *reinterpret_cast<ulonglong2 *>(rowp + g) = v;
            }
        }

        // One thread per row: the reference loop nest as written, for tiny rings (logn < 4) where a
        // radix-16 tile does not exist. Only the tests use such sizes.
        template <int DIR>
        __global__ void ntt_serial_kernel(u64 *__restrict__ data, const PrimeDev *__restrict__ primes, RowMap map,
                                          int logn, int flags, size_t nrows)
        {
            const size_t row = blockIdx.x * static_cast<size_t>(blockDim.x) + threadIdx.x;
            if (row >= nrows)
                return;
            const unsigned short pid = map.prime[row % map.rows];
            if (pid == kSkipRow)
                return;
            const PrimeDev P = primes[pid];
            const u64 p = P.p, two_p = P.two_p;
            u64 *x = data + (row << logn);
            const int N = 1 << logn;
            const bool strict = (flags & kNttStrict) != 0;
            for (int step = 0; step < logn; step++)
            {
                const int b = DIR == 0 ? logn - 1 - step : step;
                const int h = 1 << b;
                for (int j = 0; j < N; j++)
                {
                    if (j & h)
                        continue;
                    const int ti = (N + j) >> (b + 1);
                    u64 W = (DIR == 0 ? P.fwd : P.inv)[2 * ti], Ws = (DIR == 0 ? P.fwd : P.inv)[2 * ti + 1];
                    if (DIR == 0)
                    {
                        u64 u = x[j];
                        if (strict)
                            u = u >= two_p ? u - two_p : u;
                        else if (b == 0)
                            u = barrett_lazy(u, P.rdp, p);
                        u64 v = mulmod_lazy(x[j + h], W, Ws, p);
                        x[j] = u + v;
                        x[j + h] = u - v + two_p;
                    }
                    else
                    {
                        const bool top = b == logn - 1;
                        if (top)
                        {
                            W = P.inv_n_w;
                            Ws = P.inv_n_w_shoup;
                        }
                        u64 u = x[j], v = x[j + h];
                        u64 tt = u + v;
                        tt = tt >= two_p ? tt - two_p : tt;
                        if (top)
                            tt = mulmod_lazy(tt, P.inv_n, P.inv_n_shoup, p);
                        x[j] = tt;
                        x[j + h] = mulmod_lazy(u - v + two_p, W, Ws, p);
                    }
                }
            }
            if (flags & kNttCanonical)
                for (int j = 0; j < N; j++)
                {
                    u64 v = x[j];
                    if (DIR == 0)
                        v = v >= two_p ? v - two_p : v;
                    x[j] = v >= p ? v - p : v;
                }
        }

        // ----------------------------------------------------------------------------------------
        // Single-pass forward NTT for logn = 14..16 ("half-row" kernel).
        //
        // After the top butterfly layer (gap N/2) the two halves of a row are independent sub-transforms.
        // One workgroup owns one half (2^T coefficients, T = logn-1): it reads BOTH halves with 16-byte
        // coalesced loads, applies the top layer on the fly (the sibling workgroup of the other half
        // recomputes the same products: +1/logn of the multiplies, and its reads of the shared half are
        // L2 hits when the two run on the same XCD, which the block -> (row, half) map arranges), and then
        // finishes all T remaining layers on chip: 32 coefficients per lane (5 index bits), radix-16 register
        // rounds, LDS exchanges between rounds. The LDS holds only half of the tile, so every exchange runs
        // in two phases keyed on index bit 0, which stays in the registers through all rounds ("sticky"):
        // coefficients with bit0 = p only ever move between slots with slot-bit0 = p. Only the exchange between
        // arrangements 1 and 2 crosses waves (workgroup barriers); the others stay inside a wave (see kWaveLocal below).
        // HBM traffic: one read of the row (+ the sibling re-read, mostly from L2) and one write.
        //
        // Arrangement R (R = 1..3: compute rounds on index bits [T-4R, T-4R+4); R = 4: final round on the
        // remaining T-12 low bits): slot bit 0 <-> index bit 0 always; the other slot bits and the lane id
        // cover the rest as two runs of consecutive index bits.
        //
        // Round 4, wave-local arrangements (SEALHIP_NTT_WAVE_LOCAL, default on). A workgroup has 2^(T-11) waves, and in
        // arrangements 2 and 3 the wave number of a lane (tid >> 6) is exactly index bits [11, T): the lane bits below
        // cover everything else. With arrangement 4's fillers on index bits [f+6, 11) instead of the top bits -- lane bits
        // 0..5 <-> index bits [f, f+6), wave number <-> bits [11, T) there as well -- a wave owns the SAME 2048 indices
        // (1024 LDS words, one contiguous range of the exchange buffer) from arrangement 2 to the store: the exchanges
        // 2 <-> 3 and 3 <-> 4 move data between lanes of one wave only. LDS instructions of a wave execute in order, so those
        // exchanges need no s_barrier at all (h_exchange): 4 workgroup barriers per transform instead of 12, and the eight
        // waves of a workgroup drift apart after round 1 -- one wave's exchange overlaps another's arithmetic inside the same
        // workgroup, where before every wave of it waited at the same barrier. Memory side unchanged per instruction (a wave
        // still covers 2^(f+6) consecutive coefficients per final-round group); its groups are now neighbours (a contiguous
        // 16 KB per wave) instead of 16 KB apart.
#ifndef SEALHIP_NTT_FWD_DENSE
#define SEALHIP_NTT_FWD_DENSE 1 // (0: A/B build without the dense lazy forward schedule of STRICT mode's 60-bit rows)
#endif
#define SEALHIP_NTT_FWD_DENSE_DEFAULT (SEALHIP_NTT_FWD_DENSE != 0)
#ifndef SEALHIP_NTT_WAVE_LOCAL
#define SEALHIP_NTT_WAVE_LOCAL 1
#endif
        constexpr bool kWaveLocal = SEALHIP_NTT_WAVE_LOCAL != 0;
        template <int T, int R>
        struct Arr
        {
            static_assert(T >= 13 && T <= 15, "half-row shapes of N = 2^14 .. 2^16: 2^(T-5) lanes, i.e. 2^(T-11) waves <-> index bits [11, T)");
            static constexpr int beta = R <= 3 ? T - 4 * R : 0;
            static constexpr int f = T - 12; // low bits left for the final round (incl. bit 0)
            static constexpr int slot_bit(int w)
            {
                if (w == 0)
                    return 0;
                if (R <= 3)
                    return beta + (w - 1);
                if (w < f)
                    return w;
                if (kWaveLocal)
                    return f + 6 + (w - f);   // fillers: index bits [f+6, 11), just above the lane's
                return T - (5 - f) + (w - f); // fillers: the top index bits
            }
            static constexpr int low_start = R <= 3 ? 1 : f;
            static constexpr int low_len = R <= 3 ? beta - 1 : T - 5;
            static constexpr int high_start = beta + 4;
            __device__ static __forceinline__ int tid_index(int tid)
            {
                if constexpr (R == 4 && kWaveLocal)
                    return ((tid & 63) << f) | ((tid >> 6) << 11);
                int v = (tid & ((1 << low_len) - 1)) << low_start;
                if (R <= 3 && low_len < T - 5)
                    v |= (tid >> low_len) << high_start;
                return v;
            }
            // the wave number is index bits [11, T) in this arrangement (2, 3 always; 4 in the wave-local form)
            static constexpr bool wave_owned = R == 2 || R == 3 || (R == 4 && kWaveLocal);
            static constexpr int slot_index(int s)
            {
                int r = 0;
                for (int w = 0; w < 5; w++)
                    if ((s >> w) & 1)
                        r |= 1 << slot_bit(w);
                return r;
            }
            // contribution of the slot bits above w to the twiddle index of a butterfly on slot bit w
            static constexpr int tw_offset(int s, int w)
            {
                int r = 0;
                for (int v = w + 1; v < 5; v++)
                    if ((s >> v) & 1)
                        r |= 1 << (slot_bit(v) - slot_bit(w) - 1);
                return r;
            }
        };

        // streaming 16-byte store / load: the transformed rows are not read again by this kernel, keeping them out of
        // the way of the twiddle tables in L2 is worth 4 % (nontemporal hint)
        typedef u64 u64x2_nt __attribute__((ext_vector_type(2)));
        __device__ __forceinline__ void store_nt(u64 *p, u64 a, u64 b)
        {
            u64x2_nt v;
            v.x = a;
            v.y = b;
            __builtin_nontemporal_store(v, reinterpret_cast<u64x2_nt *>(p));
        }

        // LDS image of the exchange buffer: two pad words per 32 and one more for the odd 16-word blocks. With
        // 2 * (e >> 5) alone the final arrangement (lanes two words apart) had two-way bank conflicts
        // (SQ_LDS_BANK_CONFLICT 2048 per row at N = 2^15); this form measures zero (tools/hpad_sweep.sh).
        __host__ __device__ constexpr int hpad(int e)
        {
            return e + 2 * (e >> 5) + ((e >> 4) & 1);
        }

        template <int T, int RA, int RB>
        __device__ __forceinline__ void h_exchange(u64 (&x)[32], u64 *lds, int tid)
        {
            const int pa = hpad(Arr<T, RA>::tid_index(tid) >> 1);
            const int pb = hpad(Arr<T, RB>::tid_index(tid) >> 1);
            // both arrangements wave-owned: every word a wave writes is read by the same wave and by no other. The LDS
            // executes a wave's instructions in order, so the hardware needs nothing; the fences keep the compiler from
            // moving a read above the writes it depends on (they emit no instruction at wavefront scope).
            constexpr bool LOCAL = kWaveLocal && Arr<T, RA>::wave_owned && Arr<T, RB>::wave_owned;
            const auto sync = [] {
                if constexpr (LOCAL)
                {
                    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
                    __builtin_amdgcn_wave_barrier();
                    __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "wavefront");
                }
                else
                    __syncthreads();
            };
#pragma unroll
            for (int phase = 0; phase < 2; phase++)
            {
#pragma unroll
                for (int s = 0; s < 32; s++)
                    if ((s & 1) == phase)
                        lds[pa + hpad(Arr<T, RA>::slot_index(s) >> 1)] = x[s];
                sync();
#pragma unroll
                for (int s = 0; s < 32; s++)
                    if ((s & 1) == phase)
                        x[s] = lds[pb + hpad(Arr<T, RB>::slot_index(s) >> 1)];
                sync();
            }
        }

        // Twiddle tables are read-only for the lifetime of a context: load them through the global
        // (address_space 1) or, when the index is wave-uniform, the constant (address_space 4, scalar cache)
        // address space instead of the generic pointer stored in PrimeDev (which would give flat loads).
        typedef u64 u64x2 __attribute__((ext_vector_type(2)));
        typedef const __attribute__((address_space(1))) u64x2 *tw_global_t;
        typedef const __attribute__((address_space(4))) u64x2 *tw_const_t;
        // the floating-point variant's tables hold one double per twiddle (PrimeDev::fwd_d / inv_d), moved as 64-bit words
        typedef const __attribute__((address_space(1))) u64 *twd_global_t;
        typedef const __attribute__((address_space(4))) u64 *twd_const_t;
        // In the floating-point instances (STRICT == 3 forward, MODE == 2 inverse) the registers x[] hold the bit patterns
        // of doubles, the parameters named two_p / neg_p carry the bits of p and 1/p as doubles, and tw points to the
        // double table. Reduction schedule of the forward transform (ntt_bounds.hpp section 4 holds the derivation and the
        // recurrence the CPU test runs). With U = 2^50 >= p and B a bound on the magnitudes, a layer gives
        // |y*w mod p| <= (0.5 + 3 * 2^-53 |y|) p -- the quotient estimate's rounding, the relative 2^-52 of h * (1/p) AND the
        // exact rounding error l of the product -- so B' <= 1.375 B + 0.5 U: from a reduction (B = 0.5 U) FIVE layers stay
        // below 8 U = 2^53 (1.19, 2.13, 3.43, 5.22, 7.68), from raw inputs below 2^52 the top layer does (6 U). Hence: all
        // values to [-p/2, p/2] after the top layer, after round 2's first layer and after round 3's second (spans of 5, 5 and
        // at most 2 + 3 layers, the last followed by the canonicalisation): every operation is exact. (Round 2 reduced after
        // round 2's second layer and before the final round -- two spans of six layers, sound only under the bound
        // (0.5 + 2^-52 |y|) p, which ignores l: ADVICE r02.)
        __device__ __forceinline__ void fp_reduce_all(u64 (&x)[32], u64 p_bits, u64 pinv_bits)
        {
            const double p = fp_of(p_bits), pinv = fp_of(pinv_bits);
#pragma unroll
            for (int i = 0; i < 32; i++)
                x[i] = fp_bits(fp_reduce(fp_of(x[i]), p, pinv));
        }

#ifndef SEALHIP_NTT_IL
#define SEALHIP_NTT_IL 4
#endif
        constexpr int kIL = SEALHIP_NTT_IL; // butterflies advanced in lock step (devmath.hpp: butterflies_fwd_hs)

        // STRICT == 2 (approximate Shoup quotient): which form (ntt_bounds.hpp section 2: bounds::kFwdApxLevel) -- 1: round 2's
        // (hi32(y0*s0) dropped, product below 3p); 2: round 4's carry-free quotient (devmath.hpp mulhi_apx2, below 4p).
        // The butterfly's second output is u - v + (that bound), so the bound is what the layers add.
        template <int STRICT>
        constexpr int kApx = STRICT == 2 ? bounds::kFwdApxLevel : 0;
        using ZeroPairs = ZeroHi<2>; // devmath.hpp mulhi_apx2: written where a phase starts (two v_mov), each used by two of the four lock-step butterflies
        // The final round at N = 2^15 runs at the register cap (two stages of prefetched twiddles, 48 registers): four more for
        // zero-high pairs spill two coefficients. Its 32 butterflies (of 272) keep the level-1 quotient there -- a product
        // below 3p under a schedule that allows 4p, so the bounds of level 2 cover it (ntt_bounds.hpp section 2).
#ifndef SEALHIP_NTT_FINAL_ZP
#define SEALHIP_NTT_FINAL_ZP 2
#endif
        constexpr int kFinalZeroPairs = SEALHIP_NTT_FINAL_ZP;
#ifndef SEALHIP_NTT_FINAL_APX_F2
#define SEALHIP_NTT_FINAL_APX_F2 1
#endif
        template <int T, int STRICT>
        constexpr int kFinalApx = (kApx<STRICT> == 2 && (T - 12) == 2) ? SEALHIP_NTT_FINAL_APX_F2 : kApx<STRICT>;
        template <int STRICT>
        __device__ __forceinline__ u64 fwd_addend(u64 two_p, u64 neg_p)
        {
            if constexpr (STRICT != 2)
                return two_p;
            else if constexpr (bounds::kFwdApxLevel == 2)
            {
                u64 a = two_p << 1; // 4p, opaque: left visible the compiler rewrites (u << 1) + (2p << 1) as (u + 2p) << 1, two
                asm("" : "+s"(a));  // 64-bit instructions where v_lshl_add_u64 does it in one
                return a;
            }
            else
                return two_p - neg_p; // 3p
        }

        // final round, one group at a time: the low f index bits of the 2^f registers that share the filler
        // slot bits G are finished (layers f-1 .. 0) and stored right away, which bounds the live twiddles
        // Measurement-only hooks (compiled with -DSEALHIP_NTT_EXPERIMENT, driven by SEALHIP_NTT_SKIP): drop the
        // arithmetic (0x100), the LDS exchanges (0x200) or the top-layer products (0x400) to time the rest.
#ifdef SEALHIP_NTT_EXPERIMENT
#define NTT_EXP(flags, bit) (((flags) & (bit)) != 0)
#else
#define NTT_EXP(flags, bit) false
#endif

        // Final-round stores. A lane finishes runs of 2^f consecutive coefficients; for f >= 2 storing them from there
        // means 16-byte pieces at a 2^f * 8-byte stride per instruction -- every 128-byte line is written by 2^(f-1)
        // different instructions, and a kernel that does nothing but these stores reaches 2.0 TB/s at f = 2 (3.0 at
        // f = 3) against 5.3 TB/s for the contiguous stores of f = 1 (profiles/r02/ntt_store_pattern.txt). So for
        // f >= 2 the finished values take one more trip through the LDS, back to arrangement 1 (a lane holds pairs, the
        // lanes of a wave are consecutive pairs), and every store instruction writes one contiguous kilobyte.
        // Which instances take the trip (bit mask): 1 the floating-point ones, 2 the integer ones at f = 2, 4 the integer
        // ones at f = 3. Inside the pipelines the integer instances are bound by instruction issue, not by their stores:
        // at f = 2 the extra exchange costs them 2 % (config 3: 18.8 vs 18.4 ms of forward transforms per 1024 pairs)
        // although the transform alone gains 6 %; the floating-point instances gain 10 % in the config-4 key switch.
#ifndef SEALHIP_NTT_STORE_EXCHANGE
#define SEALHIP_NTT_STORE_EXCHANGE 5
#endif
        // Round 3: the transposition in registers instead. What is slow about the f >= 2 pattern is not the 16-byte pieces
        // as such but the *streaming* (nontemporal) stores of them: tools/ubench_store_pattern.hip writes the same half
        // rows at 1.9 TB/s nontemporal against 5.7 TB/s with plain stores (the L2 merges the two instructions' pieces),
        // and a nontemporal instruction is fast (5.5 TB/s) as soon as the wave as a whole covers contiguous memory --
        // which lane writes which piece does not matter (profiles/r03/store_pattern_ubench.txt). v_permlane32_swap
        // (lanes 32-63 of one register <-> lanes 0-31 of another) is exactly that transposition for f = 2: before,
        // lane (l5, r) holds pairs h = 0, 1 of its run; after swap(pair 0, pair 1) register h of lane (l5, r) holds pair
        // l5 of lane (h, r), so instruction h writes the 128 consecutive coefficients of half-wave h: one dword move per
        // dword, no LDS, no barrier. For f = 3 a v_permlane16_swap step (rows of 16 lanes) transposes the second bit.
        // Mask bits as in SEALHIP_NTT_STORE_EXCHANGE; the swap takes precedence (mode 7 keeps the trip: its store phase
        // reads the product rows at the same addresses).
#ifndef SEALHIP_NTT_STORE_SWAP
#define SEALHIP_NTT_STORE_SWAP 7
#endif
#ifndef SEALHIP_NTT_STORE_NT
#define SEALHIP_NTT_STORE_NT 1
#endif
        template <int T, int STRICT, int REDUCE = 0>
        constexpr bool kStoreSwap = (T - 12) >= 2 && REDUCE != 7 &&
                                    ((SEALHIP_NTT_STORE_SWAP) & (STRICT == 3 ? 1 : ((T - 12) == 2 ? 2 : 4))) != 0;
        template <int T, int STRICT, int REDUCE = 0>
        constexpr bool kStoreExchange =
            (T - 12) >= 2 && !kStoreSwap<T, STRICT, REDUCE> &&
            ((SEALHIP_NTT_STORE_EXCHANGE) & (STRICT == 3 ? 1 : ((T - 12) == 2 ? 2 : 4))) != 0;

        __device__ __forceinline__ void swap_half_waves(u64 &a, u64 &b) // lanes 32-63 of a <-> lanes 0-31 of b
        {
            const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(a), static_cast<unsigned>(b), false, false);
            const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(a >> 32), static_cast<unsigned>(b >> 32), false, false);
            a = lo[0] | (static_cast<u64>(hi[0]) << 32);
            b = lo[1] | (static_cast<u64>(hi[1]) << 32);
        }
        __device__ __forceinline__ void swap_rows16(u64 &a, u64 &b) // odd 16-lane rows of a <-> even rows of b
        {
            const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(a), static_cast<unsigned>(b), false, false);
            const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(a >> 32), static_cast<unsigned>(b >> 32), false, false);
            a = lo[0] | (static_cast<u64>(hi[0]) << 32);
            b = lo[1] | (static_cast<u64>(hi[1]) << 32);
        }
        // Final-round group G (2^f finished registers, runs of 2^f consecutive coefficients per lane) -> memory through the
        // register transposition: afterwards pair register i of lane L holds pair (L >> 4 or 5 bits) of the lane whose
        // those bits are i, so instruction i writes the i-th 128-coefficient piece of the wave's 2^(f+6) coefficients.
        template <int T, int G>
        __device__ __forceinline__ void h_store_group_swapped(u64 (&x)[32], u64 *__restrict__ rowp, int jb)
        {
            constexpr int f = T - 12;
            static_assert(f == 2 || f == 3, "register transposition: runs of 4 or 8 coefficients");
            constexpr int s = G << f;
            const int j = jb & ((1 << T) | ((1 << T) - 1)); // (the experiment build keeps its hooks above)
            // lane part of the address: the wave's base, then r * 2^f + (the swapped lane bits) * 2
            // (wave-local form: the lane is index bits [f, f+6), everything above -- no filler bit is set in j -- is the base)
            const int tid = kWaveLocal ? ((j >> f) & 63) : ((j & ((1 << T) - 1)) >> f);
            int base = kWaveLocal ? (j & ~((1 << (6 + f)) - 1)) : ((j & (1 << T)) + ((tid >> 6) << (6 + f)));
            if constexpr (f == 2)
            {
                base += ((tid & 31) << 2) + (((tid >> 5) & 1) << 1);
                swap_half_waves(x[s], x[s + 2]);
                swap_half_waves(x[s + 1], x[s + 3]);
            }
            else
            {
                base += ((tid & 15) << 3) + (((tid >> 5) & 1) << 2) + (((tid >> 4) & 1) << 1);
#pragma unroll
                for (int e = 0; e < 4; e++)
                    swap_half_waves(x[s + e], x[s + 4 + e]);
#pragma unroll
                for (int e = 0; e < 2; e++)
                {
                    swap_rows16(x[s + e], x[s + 2 + e]);
                    swap_rows16(x[s + 4 + e], x[s + 6 + e]);
                }
            }
            u64 *dst = rowp + base + Arr<T, 4>::slot_index(s);
#pragma unroll
            for (int i = 0; i < (1 << (f - 1)); i++)
            {
                if (SEALHIP_NTT_STORE_NT)
                    store_nt(dst + i * 128, x[s + 2 * i], x[s + 2 * i + 1]);
                else
                {
                    ulonglong2 v;
                    v.x = x[s + 2 * i];
                    v.y = x[s + 2 * i + 1];
                    *reinterpret_cast<ulonglong2 *>(dst + i * 128) = v;
                }
            }
        }

        // SX: what happens to the finished words -- 0 stored from arrangement 4 as they are, 1 kept for the LDS trip
        // (kStoreExchange), 2 transposed in registers and stored group by group (kStoreSwap)
        template <int T, int STRICT, int G, bool ROUT, int SX>
        __device__ __forceinline__ void h_final_group(u64 (&x)[32], const u64 *__restrict__ tw, u64 *__restrict__ rowp,
                                                      int jb, int N, u64 p, u64 two_p, u64 neg_p, u64 rdp, int fin, ZeroPairs &zp)
        {
            constexpr int f = T - 12;
#pragma unroll
            for (int W = f - 1; W >= 0; W--)
            {
                if (NTT_EXP(N, 0x100 << 20))
                    break;
                const int gb = Arr<T, 4>::slot_bit(W);
                const int tb = ((N & 0xFFFFF) + jb) >> (gb + 1); // (the experiment build carries its hooks in N's top bits)
                const int bit = 1 << W;
#pragma unroll
                for (int e = 0; e < (1 << f); e++)
                {
                    if (e & bit)
                        continue;
                    const int s = (G << f) | e;
                    if constexpr (STRICT == 3)
                    {
                        fp_butterfly_fwd(x[s], x[s | bit], ((twd_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)], fp_of(two_p),
                                         fp_of(neg_p));
                        continue;
                    }
                    const u64x2 Wv = ((tw_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)];
                    if (STRICT == 1)
                        x[s] = x[s] >= two_p ? x[s] - two_p : x[s];
                    else if (gb == 0 && !(fin & 2)) // fin & 2: the consumer takes any representative and nothing can wrap
                        x[s] = barrett_lazy_hs(x[s], rdp, neg_p);
                    if constexpr (kFinalApx<T, STRICT> == 2)
                        butterfly_fwd_apx2<false>(x[s], x[s | bit], Wv.x, Wv.y, neg_p, fwd_addend<STRICT>(two_p, neg_p), zp.z[(((e >> (W + 1)) << W) | (e & (bit - 1))) & (kFinalZeroPairs - 1)]);
                    else
                        butterfly_fwd_hs<false, kFinalApx<T, STRICT>>(x[s], x[s | bit], Wv.x, Wv.y, neg_p, fwd_addend<STRICT>(two_p, neg_p));
                }
            }
#pragma unroll
            for (int e = 0; e < (1 << f); e += 2)
            {
                const int s = (G << f) | e;
                ulonglong2 v;
                v.x = x[s];
                v.y = x[s + 1];
                if constexpr (STRICT == 3)
                {
                    // canonical residues always: they serve kNttCanonical and every any-representative consumer alike
                    v.x = fp_to_u64(fp_canonical(fp_of(v.x), fp_of(two_p), fp_of(neg_p)));
                    v.y = fp_to_u64(fp_canonical(fp_of(v.y), fp_of(two_p), fp_of(neg_p)));
                }
                else if (fin & 1)
                {
                    if constexpr (STRICT == 2)
                    {
                        // canonical output of the approximate-quotient schedule (values below (2 + g log n) p <= 66p,
                        // ntt_bounds.hpp section 2): one step to [0, 2p), one conditional subtraction. fin & 4: the step is
                        // the single-precision quotient estimate (rdp then carries the bits of its constant), else Barrett
                        if (fin & 4)
                        {
                            const float cq = __uint_as_float(static_cast<unsigned>(rdp));
                            v.x = reduce_small_quot(v.x, cq, neg_p);
                            v.y = reduce_small_quot(v.y, cq, neg_p);
                        }
                        else
                        {
                            v.x = barrett_lazy_hs(v.x, rdp, neg_p);
                            v.y = barrett_lazy_hs(v.y, rdp, neg_p);
                        }
                    }
                    else
                    {
                        v.x = v.x >= two_p ? v.x - two_p : v.x;
                        v.y = v.y >= two_p ? v.y - two_p : v.y;
                    }
                    v.x = v.x >= p ? v.x - p : v.x;
                    v.y = v.y >= p ? v.y - p : v.y;
                }
                else if constexpr (ROUT && STRICT == 4)
                {
                    // dense lazy schedule: words below 16p -> [0, 2p) (rdp carries the bits of the quotient constant)
                    const float cq = __uint_as_float(static_cast<unsigned>(rdp));
                    v.x = reduce_small_quot(v.x, cq, neg_p);
                    v.y = reduce_small_quot(v.y, cq, neg_p);
                }
                else if constexpr (ROUT)
                {
                    // kNttReduceOut: [0, 4p) -> [0, 2p), same residue. (The last layer reduces its first operand and its
                    // product below 2p before it adds them -- ForwardLazyLast, ntt.cpp:254-261 -- so even on the 60-bit rows,
                    // where earlier layers wrap, what it outputs is below 4p.)
                    v.x = v.x >= two_p ? v.x - two_p : v.x;
                    v.y = v.y >= two_p ? v.y - two_p : v.y;
                }
                if (NTT_EXP(N, 0x800 << 20) && v.x != 0x1234567)
                    continue;
                if constexpr (SX != 0)
                {
                    x[s] = v.x; // stored by h_store_rows after the trip back to arrangement 1, or transposed below
                    x[s + 1] = v.y;
                }
                else
                    // (plain store: a lane's 64-byte run is written by four instructions and the L2 has to merge them;
                    //  streaming stores cost 12 % there)
                    *reinterpret_cast<ulonglong2 *>(rowp + (jb & ((1 << T) | ((1 << T) - 1))) + Arr<T, 4>::slot_index(s)) = v;
            }
            if constexpr (SX == 2)
                if (!NTT_EXP(N, 0x800 << 20))
                    h_store_group_swapped<T, G>(x, rowp, jb);
        }

        template <int T, int STRICT, int G, int NG, bool ROUT, int SX>
        struct FinalGroups
        {
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64 *__restrict__ tw, u64 *__restrict__ rowp,
                                                       int jb, int N, u64 p, u64 two_p, u64 neg_p, u64 rdp, int fin, ZeroPairs &zp)
            {
                h_final_group<T, STRICT, G, ROUT, SX>(x, tw, rowp, jb, N, p, two_p, neg_p, rdp, fin, zp);
                if ((G & 1) == 1)
                    __builtin_amdgcn_sched_barrier(0); // keep the compiler from hoisting every group's twiddle loads
                FinalGroups<T, STRICT, G + 1, NG, ROUT, SX>::run(x, tw, rowp, jb, N, p, two_p, neg_p, rdp, fin, zp);
            }
        };
        template <int T, int STRICT, int NG, bool ROUT, int SX>
        struct FinalGroups<T, STRICT, NG, NG, ROUT, SX>
        {
            __device__ static __forceinline__ void run(u64 (&)[32], const u64 *, u64 *, int, int, u64, u64, u64, u64, int, ZeroPairs &)
            {}
        };

        // ---- pipelined final round. A stage = SG groups; the twiddles of stage k+1 are requested before stage k is
        // computed and stored, so their L2 latency is covered by a stage of arithmetic instead of being exposed at
        // every sched_barrier. Twiddles of one group, in the order used: layer W = f-1 (1 entry), f-2 (2), ... 0.
        template <int T>
        struct FinalStage
        {
            static constexpr int f = T - 12;
            static constexpr int NTW = (1 << f) - 1;            // twiddles per group
            static constexpr int SG = f == 1 ? 4 : (f == 2 ? 2 : 1); // groups per stage
            static constexpr int NG = 1 << (5 - f);
            static constexpr int NS = NG / SG;
            static constexpr bool PIPE = f <= 2; // f = 3: two stages of 28 twiddle registers do not fit
        };

        template <int T, int G, bool FP = false>
        __device__ __forceinline__ void h_final_tw(u64x2 *tg, const u64 *__restrict__ tw, int jb, int N)
        {
            constexpr int f = T - 12;
#pragma unroll
            for (int W = f - 1; W >= 0; W--)
            {
                const int tb = ((N & 0xFFFFF) + jb) >> (Arr<T, 4>::slot_bit(W) + 1);
#pragma unroll
                for (int o = 0; o < (1 << (f - 1 - W)); o++)
                {
                    const int s = (G << f) | (o << (W + 1));
                    if constexpr (FP)
                        tg[(1 << (f - 1 - W)) - 1 + o].x = ((twd_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)];
                    else
                        tg[(1 << (f - 1 - W)) - 1 + o] = ((tw_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)];
                }
            }
        }

        template <int T, int STRICT, int G, bool ROUT, int SX>
        __device__ __forceinline__ void h_final_group_regs(u64 (&x)[32], const u64x2 *tg, u64 *__restrict__ rowp, int jb,
                                                           int N, u64 p, u64 two_p, u64 neg_p, u64 rdp, int fin, ZeroPairs &zp)
        {
            constexpr int f = T - 12;
#pragma unroll
            for (int W = f - 1; W >= 0; W--)
            {
                if (NTT_EXP(N, 0x100 << 20))
                    break;
                const int gb = Arr<T, 4>::slot_bit(W);
                const int bit = 1 << W;
#pragma unroll
                for (int e = 0; e < (1 << f); e++)
                {
                    if (e & bit)
                        continue;
                    const int s = (G << f) | e;
                    const u64x2 Wv = tg[(1 << (f - 1 - W)) - 1 + (e >> (W + 1))];
                    if constexpr (STRICT == 3)
                    {
                        fp_butterfly_fwd(x[s], x[s | bit], Wv.x, fp_of(two_p), fp_of(neg_p));
                        continue;
                    }
                    if (STRICT == 1)
                        x[s] = x[s] >= two_p ? x[s] - two_p : x[s];
                    else if (gb == 0 && !(fin & 2)) // fin & 2: the consumer takes any representative and nothing can wrap
                        x[s] = barrett_lazy_hs(x[s], rdp, neg_p);
                    if constexpr (kFinalApx<T, STRICT> == 2)
                        butterfly_fwd_apx2<false>(x[s], x[s | bit], Wv.x, Wv.y, neg_p, fwd_addend<STRICT>(two_p, neg_p), zp.z[(((e >> (W + 1)) << W) | (e & (bit - 1))) & (kFinalZeroPairs - 1)]);
                    else
                        butterfly_fwd_hs<false, kFinalApx<T, STRICT>>(x[s], x[s | bit], Wv.x, Wv.y, neg_p, fwd_addend<STRICT>(two_p, neg_p));
                }
            }
#pragma unroll
            for (int e = 0; e < (1 << f); e += 2)
            {
                const int s = (G << f) | e;
                ulonglong2 v;
                v.x = x[s];
                v.y = x[s + 1];
                if constexpr (STRICT == 3)
                {
                    // canonical residues always: they serve kNttCanonical and every any-representative consumer alike
                    v.x = fp_to_u64(fp_canonical(fp_of(v.x), fp_of(two_p), fp_of(neg_p)));
                    v.y = fp_to_u64(fp_canonical(fp_of(v.y), fp_of(two_p), fp_of(neg_p)));
                }
                else if (fin & 1)
                {
                    if constexpr (STRICT == 2)
                    {
                        // canonical output of the approximate-quotient schedule (values below (2 + g log n) p <= 66p,
                        // ntt_bounds.hpp section 2): one step to [0, 2p), one conditional subtraction. fin & 4: the step is
                        // the single-precision quotient estimate (rdp then carries the bits of its constant), else Barrett
                        if (fin & 4)
                        {
                            const float cq = __uint_as_float(static_cast<unsigned>(rdp));
                            v.x = reduce_small_quot(v.x, cq, neg_p);
                            v.y = reduce_small_quot(v.y, cq, neg_p);
                        }
                        else
                        {
                            v.x = barrett_lazy_hs(v.x, rdp, neg_p);
                            v.y = barrett_lazy_hs(v.y, rdp, neg_p);
                        }
                    }
                    else
                    {
                        v.x = v.x >= two_p ? v.x - two_p : v.x;
                        v.y = v.y >= two_p ? v.y - two_p : v.y;
                    }
                    v.x = v.x >= p ? v.x - p : v.x;
                    v.y = v.y >= p ? v.y - p : v.y;
                }
                else if constexpr (ROUT && STRICT == 4)
                {
                    // dense lazy schedule: words below 16p -> [0, 2p) (rdp carries the bits of the quotient constant)
                    const float cq = __uint_as_float(static_cast<unsigned>(rdp));
                    v.x = reduce_small_quot(v.x, cq, neg_p);
                    v.y = reduce_small_quot(v.y, cq, neg_p);
                }
                else if constexpr (ROUT)
                {
                    // kNttReduceOut: [0, 4p) -> [0, 2p), same residue. (The last layer reduces its first operand and its
                    // product below 2p before it adds them -- ForwardLazyLast, ntt.cpp:254-261 -- so even on the 60-bit rows,
                    // where earlier layers wrap, what it outputs is below 4p.)
                    v.x = v.x >= two_p ? v.x - two_p : v.x;
                    v.y = v.y >= two_p ? v.y - two_p : v.y;
                }
                if (NTT_EXP(N, 0x800 << 20) && v.x != 0x1234567)
                    continue;
                if constexpr (SX != 0)
                {
                    x[s] = v.x; // stored by h_store_rows after the trip back to arrangement 1, or transposed below
                    x[s + 1] = v.y;
                }
                else if (SEALHIP_NTT_STORE_NT)
                    store_nt(rowp + (jb & ((1 << T) | ((1 << T) - 1))) + Arr<T, 4>::slot_index(s), v.x, v.y);
                else
                    *reinterpret_cast<ulonglong2 *>(rowp + (jb & ((1 << T) | ((1 << T) - 1))) + Arr<T, 4>::slot_index(s)) = v;
            }
            if constexpr (SX == 2)
                if (!NTT_EXP(N, 0x800 << 20))
                    h_store_group_swapped<T, G>(x, rowp, jb);
        }

        template <int T, int ST, int I = 0, bool FP = false>
        struct StageTw // twiddle loads of stage ST
        {
            __device__ static __forceinline__ void load(u64x2 *tg, const u64 *__restrict__ tw, int jb, int N)
            {
                h_final_tw<T, ST * FinalStage<T>::SG + I, FP>(tg + I * FinalStage<T>::NTW, tw, jb, N);
                if constexpr (I + 1 < FinalStage<T>::SG)
                    StageTw<T, ST, I + 1, FP>::load(tg, tw, jb, N);
            }
        };
        template <int T, int STRICT, bool ROUT, int SX, int ST, int I = 0>
        struct StageRun
        {
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64x2 *tg, u64 *__restrict__ rowp, int jb, int N,
                                                       u64 p, u64 two_p, u64 neg_p, u64 rdp, int fin, ZeroPairs &zp)
            {
                h_final_group_regs<T, STRICT, ST * FinalStage<T>::SG + I, ROUT, SX>(x, tg + I * FinalStage<T>::NTW, rowp, jb, N, p,
                                                                               two_p, neg_p, rdp, fin, zp);
                if constexpr (I + 1 < FinalStage<T>::SG)
                    StageRun<T, STRICT, ROUT, SX, ST, I + 1>::run(x, tg, rowp, jb, N, p, two_p, neg_p, rdp, fin, zp);
            }
        };
        template <int T, int STRICT, bool ROUT, int SX, int ST>
        struct FinalPipe
        {
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64x2 *cur, const u64 *__restrict__ tw,
                                                       u64 *__restrict__ rowp, int jb, int N, u64 p, u64 two_p, u64 neg_p,
                                                       u64 rdp, int fin, ZeroPairs &zp)
            {
                u64x2 next[FinalStage<T>::SG * FinalStage<T>::NTW];
                if constexpr (ST + 1 < FinalStage<T>::NS)
                    StageTw<T, ST + 1, 0, STRICT == 3>::load(next, tw, jb, N);
                __builtin_amdgcn_sched_barrier(0);
                StageRun<T, STRICT, ROUT, SX, ST>::run(x, cur, rowp, jb, N, p, two_p, neg_p, rdp, fin, zp);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST + 1 < FinalStage<T>::NS)
                    FinalPipe<T, STRICT, ROUT, SX, ST + 1>::run(x, next, tw, rowp, jb, N, p, two_p, neg_p, rdp, fin, zp);
            }
        };

        // ---- a compute round as a pipeline of stages. Stage K = kIL butterflies of one layer (layers W = 4, 3, 2, 1,
        // 16 / kIL stages each). The twiddles of stage K+1 are requested before stage K is computed (pinned with
        // sched_barrier), the first stage's before the preceding LDS exchange: no twiddle latency is exposed.
        template <int T, int R, int STRICT, bool UNIFORM, int K>
        struct RoundStage
        {
            static constexpr int PER = 16 / kIL;
            static constexpr int W = 4 - K / PER;
            static constexpr int C = (K % PER) * kIL;
            static constexpr int bit = 1 << W;
            static constexpr int slot(int j)
            {
                return (((C + j) >> W) << (W + 1)) | ((C + j) & (bit - 1)); // the (C+j)-th slot with bit W clear
            }
            __device__ static __forceinline__ void load(u64 (&w)[kIL], u64 (&ws)[kIL], const u64 *__restrict__ tw, int jb,
                                                        int N)
            {
                const int tb = (N + jb) >> (Arr<T, R>::slot_bit(W) + 1);
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    u64x2 Wv;
                    if constexpr (STRICT == 3)
                    {
                        Wv.x = UNIFORM ? ((twd_const_t)tw)[__builtin_amdgcn_readfirstlane(tb) + Arr<T, R>::tw_offset(slot(j), W)]
                                       : ((twd_global_t)tw)[tb + Arr<T, R>::tw_offset(slot(j), W)];
                        Wv.y = 0;
                    }
                    else if (UNIFORM)
                        Wv = ((tw_const_t)tw)[__builtin_amdgcn_readfirstlane(tb) + Arr<T, R>::tw_offset(slot(j), W)];
                    else
                        Wv = ((tw_global_t)tw)[tb + Arr<T, R>::tw_offset(slot(j), W)];
                    w[j] = Wv.x;
                    ws[j] = Wv.y;
                }
            }
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64 (&w)[kIL], const u64 (&ws)[kIL], u64 two_p,
                                                       u64 neg_p, ZeroPairs &zp)
            {
                if constexpr (STRICT == 3)
                {
#pragma unroll
                    for (int j = 0; j < kIL; j++)
                        fp_butterfly_fwd(x[slot(j)], x[slot(j) | bit], w[j], fp_of(two_p), fp_of(neg_p));
                    return;
                }
                u64 u[kIL], y[kIL];
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    u[j] = x[slot(j)];
                    y[j] = x[slot(j) | bit];
                    if (STRICT == 1)
                        u[j] = u[j] >= two_p ? u[j] - two_p : u[j];
                }
                if constexpr (kApx<STRICT> == 2)
                {
                    static_assert(kIL == 4, "two zero-high pairs for four lock-step butterflies");
                    butterflies_fwd_apx2<UNIFORM, kIL>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p), zp.z);
                }
                else
                    butterflies_fwd_hs<UNIFORM, kIL, kApx<STRICT>>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p)); // ForwardLazy, ntt.cpp:245-252
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    x[slot(j)] = u[j];
                    x[slot(j) | bit] = y[j];
                }
            }
        };
        template <int T, int R, int STRICT, bool UNIFORM, int K = 0>
        struct RoundPipe
        {
            static constexpr int NST = 4 * (16 / kIL);
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64 (&w)[kIL], const u64 (&ws)[kIL],
                                                       const u64 *__restrict__ tw, int jb, int N, u64 two_p, u64 neg_p, ZeroPairs &zp)
            {
                u64 wn[kIL], wsn[kIL];
                if constexpr (K + 1 < NST)
                    RoundStage<T, R, STRICT, UNIFORM, K + 1>::load(wn, wsn, tw, jb, N);
                __builtin_amdgcn_sched_barrier(0);
                // (before on-chip layers 5 and 10 -- after round 2's first and round 3's second layer: see fp_reduce_all)
                if constexpr (STRICT == 3 && K % (16 / kIL) == 0 && bounds::fp_fwd_reduce_before_layer(4 * (R - 1) + K / (16 / kIL)))
                    fp_reduce_all(x, two_p, neg_p);
                // dense lazy schedule (STRICT == 4, ntt_bounds.hpp section 2b): every word back below 2p before rounds 2 and 3
                if constexpr (STRICT == 4 && K == 0 && bounds::fwd_dense_reduce_before_round(R))
                {
                    const float cq = small_quot_const(0 - neg_p);
#pragma unroll
                    for (int i = 0; i < 32; i++)
                        x[i] = reduce_small_quot(x[i], cq, neg_p);
                }
                RoundStage<T, R, STRICT, UNIFORM, K>::run(x, w, ws, two_p, neg_p, zp);
                if constexpr (K + 1 < NST)
                    RoundPipe<T, R, STRICT, UNIFORM, K + 1>::run(x, wn, wsn, tw, jb, N, two_p, neg_p, zp);
            }
        };

#ifndef SEALHIP_NTT_LOAD_BATCH
#define SEALHIP_NTT_LOAD_BATCH 4
#endif
        constexpr int kLoadBatch = SEALHIP_NTT_LOAD_BATCH; // (lo, hi) 16-byte pairs per lane in flight during the load phase
        template <int T, int STRICT, int HALF, int REDUCE>
        __device__ __forceinline__ void h_load_top(u64 (&x)[32], const u64 *__restrict__ rowp,
                                                   const u64 *__restrict__ tw, int tid, u64 two_p, u64 neg_p, u64 cr1,
                                                   u64 aux_p = 0, u64 aux_cr1 = 0, const u64 *aux_top = nullptr)
        {
            const int jb = Arr<T, 1>::tid_index(tid);
            ZeroPairs zp;
            if constexpr (kApx<STRICT> == 2)
                zp.init();
            u64x2 W1;
            if constexpr (STRICT == 3)
                W1.x = ((twd_const_t)tw)[1];
            else
                W1 = ((tw_const_t)tw)[1];
#pragma unroll
            for (int batch = 0; batch < 16 / kLoadBatch; batch++)
            {
                ulonglong2 lo[kLoadBatch], hi[kLoadBatch];
#pragma unroll
                for (int i = 0; i < kLoadBatch; i++)
                {
                    const int s = (batch * kLoadBatch + i) * 2;
                    const int idx = jb + Arr<T, 1>::slot_index(s);
                    lo[i] = *reinterpret_cast<const ulonglong2 *>(rowp + idx);
                    hi[i] = *reinterpret_cast<const ulonglong2 *>(rowp + (1 << T) + idx);
                }
                if constexpr (REDUCE == 5 || REDUCE == 7)
                {
                    // mode 4 on a source row whose top inverse layer was left to us: the pair (lo, hi) = (c, c + N/2) first
                    // goes through BackwardLazyLast w.r.t. the special prime P (inputs below 2P), then -(. mod P)
#pragma unroll
                    for (int i = 0; i < kLoadBatch; i++)
                    {
                        const auto top = [&](u64 &u, u64 &v) {
                            if constexpr (STRICT == 3)
                            {
                                // aux_p / aux_cr1: P and 1/P as doubles; aux_top[0], [2]: n^-1 and w n^-1 as doubles
                                const double P = fp_of(aux_p), Pinv = fp_of(aux_cr1), ud = fp_from_u64(u), vd = fp_from_u64(v);
                                const double a0 = fp_canonical(fp_mulmod(ud + vd, fp_of(aux_top[0]), P, Pinv), P, Pinv);
                                const double a1 = fp_canonical(fp_mulmod(ud - vd, fp_of(aux_top[2]), P, Pinv), P, Pinv);
                                u = fp_bits(a0 != 0.0 ? P - a0 : 0.0);
                                v = fp_bits(a1 != 0.0 ? P - a1 : 0.0);
                            }
                            else
                            {
                                const u64 two_P = aux_p << 1;
                                u64 tt = u + v;
                                tt = tt >= two_P ? tt - two_P : tt;
                                u64 a0 = mulmod_lazy(tt, aux_top[0], aux_top[1], aux_p); // below 2P
                                u64 a1 = mulmod_lazy(u - v + two_P, aux_top[2], aux_top[3], aux_p);
                                a0 = a0 >= aux_p ? a0 - aux_p : a0;
                                a1 = a1 >= aux_p ? a1 - aux_p : a1;
                                u = a0 ? aux_p - a0 : 0;
                                v = a1 ? aux_p - a1 : 0;
                            }
                        };
                        top(lo[i].x, hi[i].x);
                        top(lo[i].y, hi[i].y);
                    }
                }
                if constexpr (REDUCE == 4)
                {
                    // CKKS mod-down with one special prime P (multi_special_primes.cpp:262-273): the word is a lazy value of
                    // the special row; the row being transformed holds (-(s mod P)) mod q. -(s mod P) is formed here as the
                    // integer P - r (0 for r = 0), which is below P < 2q: the lazy transform takes it as it is, so the
                    // separate pass that wrote these k rows and the read of them are gone.
#pragma unroll
                    for (int i = 0; i < kLoadBatch; i++)
                    {
                        const auto red = [&](u64 v) {
                            if constexpr (STRICT == 3)
                            {
                                // floating-point instance: aux_p / aux_cr1 carry P and 1/P as doubles, v < 2^52; the word
                                // stays a double (the top layer below does not convert it again)
                                const double P = fp_of(aux_p), r = fp_canonical(fp_from_u64(v), P, fp_of(aux_cr1));
                                return fp_bits(r != 0.0 ? P - r : 0.0);
                            }
                            const u64 r = barrett_reduce_63(v, aux_p, aux_cr1);
                            return r ? aux_p - r : 0;
                        };
                        lo[i].x = red(lo[i].x);
                        lo[i].y = red(lo[i].y);
                        hi[i].x = red(hi[i].x);
                        hi[i].y = red(hi[i].y);
                    }
                }
                if constexpr (REDUCE == 1 || REDUCE == 2) // gathered single-prime mod-up (multi_special_primes.cpp:103-107)
                {
                    const u64 p = STRICT == 3 ? static_cast<u64>(fp_of(two_p)) : 0 - neg_p;
                    const auto red = [&](u64 v) {
                        if constexpr (REDUCE == 2)
                            return v >= p ? v - p : v; // source prime < 2p: the canonical residue is v or v - p
                        else
                            return barrett_reduce_63(v, p, cr1);
                    };
#pragma unroll
                    for (int i = 0; i < kLoadBatch; i++)
                    {
                        lo[i].x = red(lo[i].x);
                        lo[i].y = red(lo[i].y);
                        hi[i].x = red(hi[i].x);
                        hi[i].y = red(hi[i].y);
                    }
                }
#pragma unroll
                for (int i = 0; i < kLoadBatch; i += 2)
                {
                    // four butterflies in lock step (program-ordered asm: the products stay inside their batch instead
                    // of being sunk below the loads of the later batches, which used to spill loaded values)
                    const int s = (batch * kLoadBatch + i) * 2;
                    u64 u[4] = {lo[i].x, lo[i].y, lo[i + 1].x, lo[i + 1].y};
                    u64 y[4] = {hi[i].x, hi[i].y, hi[i + 1].x, hi[i + 1].y};
                    if constexpr (STRICT == 3)
                    {
                        // inputs below 2^52 (launch_half: residues, lazy gathered values, or the treatments above)
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            if constexpr (REDUCE != 4 && REDUCE != 5 && REDUCE != 7)
                            {
                                u[j] = fp_bits(fp_from_u64(u[j]));
                                y[j] = fp_bits(fp_from_u64(y[j]));
                            }
                            fp_butterfly_fwd(u[j], y[j], W1.x, fp_of(two_p), fp_of(neg_p));
                            x[s + j] = HALF ? y[j] : u[j];
                        }
                        continue;
                    }
                    const u64 w[4] = {W1.x, W1.x, W1.x, W1.x}, ws[4] = {W1.y, W1.y, W1.y, W1.y};
                    if (STRICT == 1)
                    {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                            u[j] = u[j] >= two_p ? u[j] - two_p : u[j];
                    }
                    if constexpr (kApx<STRICT> == 2)
                        butterflies_fwd_apx2<true, 4>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p), zp.z);
                    else
                        butterflies_fwd_hs<true, 4, kApx<STRICT>>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p));
#pragma unroll
                    for (int j = 0; j < 4; j++)
                        x[s + j] = HALF ? y[j] : u[j];
                }
                // keep the next batch's loads from being hoisted over this batch's products: that costs registers
                // (spilled loaded values came back as HBM write traffic) and buys nothing (the load phase is bound by
                // the CU's load path, not by latency: tools/ntt_phases.sh)
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // ---- round 4 experiment (SEALHIP_NTT_XCHG_TOP, default 0): the top layer SHARED by the two workgroups of a row instead
        // of computed by both. Workgroup h owns the butterflies j whose index bit T-1 equals h (slot bit 4 of arrangement 1:
        // 16 of a lane's 32 slots): it loads both inputs of those, computes both outputs, keeps the one of its half and PUBLISHES
        // the other into the row at the position of the input it has just consumed (in place: position (1-h) 2^T + j is an
        // input of this very butterfly and of no other); then it signals, waits for the sibling's wave with the same lanes, and
        // reads what that wave published for it (positions h 2^T + j', bit T-1 of j' = 1-h). 16 full butterflies per lane instead
        // of 32 products with one output each. Published words and the signal cross CUs: relaxed atomics at agent scope (sc1:
        // write-through / L1-bypassing accesses), the signal ordered after the stores by a wait for their completion. The final
        // stores of a workgroup only touch positions whose last reader it is itself, so the "both have finished reading" hand-off
        // before the store phase is not needed in this form.
        // MEASURED (profiles/r04/sibling_top_exchange_ab.txt; bit-exact in both forms): standalone forward transform at N = 2^15
        // 40.3 % of the HBM roofline without it, 34.2 % with it as written here, 39.0 % with plain 16-byte stores and
        // nontemporal loads (which is only correct while the two workgroups share an L2); FP64 instances 50.9 / 40.2 / 47.8 %.
        // The 160 instructions per lane it saves cost more in the load phase (write-through stores, L1-bypassing loads, the
        // wait for the sibling's wave) than they are worth: left in the tree as the record of that, compiled out.
#ifndef SEALHIP_NTT_XCHG_TOP
#define SEALHIP_NTT_XCHG_TOP 0
#endif
        template <int T, int STRICT, int HALF, int REDUCE>
        __device__ __forceinline__ void h_load_top_xchg(u64 (&x)[32], const u64 *srcp, u64 *rowp,
                                                        const u64 *__restrict__ tw, int tid, u64 two_p, u64 neg_p, u64 cr1,
                                                        unsigned *flagw, unsigned *timeout_flag, unsigned spin_limit, bool signal)
        {
            static_assert(REDUCE <= 3, "plain or single-prime mod-up loads");
            const int jb = Arr<T, 1>::tid_index(tid);
            ZeroPairs zp;
            if constexpr (kApx<STRICT> == 2)
                zp.init();
            u64x2 W1;
            if constexpr (STRICT == 3)
                W1.x = ((twd_const_t)tw)[1];
            else
                W1 = ((tw_const_t)tw)[1];
            u64 *pub = rowp + ((1 - HALF) << T) + jb; // where the outputs of the other half go
#pragma unroll
            for (int batch = 0; batch < 2; batch++)
            {
                ulonglong2 lo[4], hi[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int s = (HALF << 4) | ((batch * 4 + i) * 2);
                    const int idx = jb + Arr<T, 1>::slot_index(s);
                    lo[i] = *reinterpret_cast<const ulonglong2 *>(srcp + idx);
                    hi[i] = *reinterpret_cast<const ulonglong2 *>(srcp + (1 << T) + idx);
                }
                if constexpr (REDUCE == 1 || REDUCE == 2) // gathered single-prime mod-up (multi_special_primes.cpp:103-107)
                {
                    const u64 p = STRICT == 3 ? static_cast<u64>(fp_of(two_p)) : 0 - neg_p;
                    const auto red = [&](u64 v) {
                        if constexpr (REDUCE == 2)
                            return v >= p ? v - p : v;
                        else
                            return barrett_reduce_63(v, p, cr1);
                    };
#pragma unroll
                    for (int i = 0; i < 4; i++)
                    {
                        lo[i].x = red(lo[i].x);
                        lo[i].y = red(lo[i].y);
                        hi[i].x = red(hi[i].x);
                        hi[i].y = red(hi[i].y);
                    }
                }
#pragma unroll
                for (int i = 0; i < 4; i += 2)
                {
                    const int s = (HALF << 4) | ((batch * 4 + i) * 2);
                    u64 u[4] = {lo[i].x, lo[i].y, lo[i + 1].x, lo[i + 1].y};
                    u64 y[4] = {hi[i].x, hi[i].y, hi[i + 1].x, hi[i + 1].y};
                    if constexpr (STRICT == 3)
                    {
#pragma unroll
                        for (int j = 0; j < 4; j++)
                        {
                            u[j] = fp_bits(fp_from_u64(u[j]));
                            y[j] = fp_bits(fp_from_u64(y[j]));
                            fp_butterfly_fwd(u[j], y[j], W1.x, fp_of(two_p), fp_of(neg_p));
                        }
                    }
                    else
                    {
                        const u64 w[4] = {W1.x, W1.x, W1.x, W1.x}, ws[4] = {W1.y, W1.y, W1.y, W1.y};
                        if (STRICT == 1)
                        {
#pragma unroll
                            for (int j = 0; j < 4; j++)
                                u[j] = u[j] >= two_p ? u[j] - two_p : u[j];
                        }
                        if constexpr (kApx<STRICT> == 2)
                            butterflies_fwd_apx2<true, 4>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p), zp.z);
                        else
                            butterflies_fwd_hs<true, 4, kApx<STRICT>>(u, y, w, ws, neg_p, fwd_addend<STRICT>(two_p, neg_p));
                    }
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        x[s + j] = HALF ? y[j] : u[j];
                        __hip_atomic_store(pub + Arr<T, 1>::slot_index(s + j), HALF ? u[j] : y[j], __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_AGENT);
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
            // the published words of this wave have left (a completed sc1 store is visible at agent scope); then the signal
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory"); // (explicit: a workgroup-scope release need not wait for stores)
            if ((tid & 63) == 0)
            {
                if (signal)
                    __hip_atomic_fetch_or(flagw, 1u << HALF, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                unsigned spins = 0;
                while (((__hip_atomic_load(flagw, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) >> (1 - HALF)) & 1u) == 0)
                {
                    __builtin_amdgcn_s_sleep(4);
                    if (++spins > spin_limit)
                    {
                        // (never observed: do not hang the device; the launch is flagged as failed for every host-visible
                        //  synchronisation point, like the hand-off of the other form)
                        __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
            const u64 *got = rowp + (HALF << T) + jb;
#pragma unroll
            for (int t = 0; t < 16; t++)
            {
                const int s = ((1 - HALF) << 4) | t;
                x[s] = __hip_atomic_load(got + Arr<T, 1>::slot_index(s), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }

        // XCD-aware block -> (row, half) map (speed only; any placement gives the same result). Blocks are dealt
        // round-robin over the 8 XCDs, each with its own 4 MB L2. Rows are enumerated prime-major
        // (v = position * npolys + poly, positions = the LIVE slots of a polynomial in slot order) and XCD x gets the
        // contiguous range [x*chunk, (x+1)*chunk): it then touches at most two primes, so their twiddle tables (512 KB
        // each at N=2^15) stay L2-resident instead of all rows_per_poly tables thrashing every L2 (measured: HBM reads
        // 583 -> 377 KB per row at N=2^15). Slots mapped to kSkipRow are left out of the enumeration: counting them
        // left the XCD that owned a skipped slot idle (the key-switch launches skip one slot in k+1, and the CKKS
        // special-row launches transform one slot in k+1: one XCD did all the work).
        // The two halves of a row are consecutive blocks of one XCD.
        struct LiveSlots
        {
            int n;
            unsigned short slot[kMaxRows];
        };
        inline LiveSlots live_slots(const RowMap &map)
        {
            LiveSlots ls{};
            for (int r = 0; r < map.rows; r++)
                if (map.prime[r] != kSkipRow)
                    ls.slot[ls.n++] = static_cast<unsigned short>(r);
            return ls;
        }
        // -> false when the block has nothing to do; else the polynomial index and the position among the live slots
        // poly_major = g > 0 (the key switch's digit launches, whose live rows all gather ONE source row): groups of g live
        // positions interleaved item by item, so that g readers of a source row run next to each other on one XCD and the row is
        // fetched from HBM n_live / g times instead of n_live times -- at the price of an XCD touching up to 2g twiddle tables
        // instead of two. Round 3, config 3 (integer instances, seven readers per source row, 512 KB per table): forward
        // transforms 17.34 ms per 1024 pairs at g = 1, 17.18 / 17.10 / 16.98 at g = 2 / 3 / 4, 17.13 at g = 7 (the tables start
        // to thrash the 4 MB L2); step 44.98 -> 44.51 ms at g = 4. Config 4 (FP64, eleven readers): the transforms gain 2.5 % at
        // g = 11 but the inner product, which reads what they wrote in the old order, loses as much: off there.
        // SEALHIP_NTT_POLY_MAJOR (bit 0: FP64 launches, bit 1: integer; default 2) and SEALHIP_NTT_POLY_GROUP (default 4).
        __device__ __forceinline__ bool half_block_map(unsigned bid, std::size_t npolys, int n_live, std::size_t chunk,
                                                       std::size_t &poly, int &position, int &half, int poly_major = 0)
        {
            const unsigned xcd = bid & 7u;
            const std::size_t slot = bid >> 3;
            half = static_cast<int>(slot & 1);
            const std::size_t v = static_cast<std::size_t>(xcd) * chunk + (slot >> 1);
            if ((slot >> 1) >= chunk || v >= npolys * static_cast<std::size_t>(n_live))
                return false;
            if (poly_major)
            {
                // groups of g consecutive live positions interleaved item by item: v = (group * npolys + poly) * g + b
                // (g = n_live: item-major; the last group may be shorter)
                const int g = poly_major;
                const std::size_t full = static_cast<std::size_t>(n_live / g) * npolys * g; // rows in complete groups
                if (v < full)
                {
                    const std::size_t gp = v / g;
                    position = static_cast<int>(gp / npolys) * g + static_cast<int>(v - gp * g);
                    poly = gp % npolys;
                }
                else
                {
                    const int rem = n_live % g; // > 0 here
                    const std::size_t u = v - full, gp = u / rem;
                    position = (n_live / g) * g + static_cast<int>(u - gp * rem);
                    poly = gp;
                }
                return true;
            }
            poly = v % npolys;
            position = static_cast<int>(v / npolys);
            return true;
        }

#ifdef SEALHIP_NTT_EXPERIMENT
        __device__ unsigned long long *g_ntt_trace = nullptr; // [block][8] timestamps (wall clock, 100 MHz)
#define NTT_STAMP(i)                                                                  \
    do                                                                                \
    {                                                                                 \
        if ((flags & 0x2000) && tid == 0 && g_ntt_trace)                              \
            g_ntt_trace[static_cast<std::size_t>(blockIdx.x) * 8 + (i)] = wall_clock64(); \
    } while (0)
#else
#define NTT_STAMP(i) \
    do               \
    {                \
    } while (0)
#endif

        // An opaque copy of the thread index. Everything a phase derives from it (coefficient indices, LDS addresses,
        // twiddle indices) is then computed where the phase starts instead of at the top of the kernel, where it
        // would sit in registers -- or in scratch, whose write-back showed up as +28 % HBM write traffic -- until used.
        __device__ __forceinline__ int fresh(int v)
        {
            asm volatile("" : "+v"(v));
            return v;
        }
        // Round 3: the thread index itself is not kept either. These kernels run at the 128-register cap, and the one value
        // every phase needs -- threadIdx.x -- was what the allocator spilled (1-6 dwords of scratch in eight inverse instances).
        // The wave's base index is uniform (an SGPR), the lane id comes from v_mbcnt: two full-rate instructions wherever a
        // phase starts, no register held across the rounds. (volatile: never merged with an earlier copy)
        __device__ __forceinline__ int fresh_tid(int wave_base)
        {
            int lane;
            asm volatile("v_mbcnt_lo_u32_b32 %0, -1, 0\n\tv_mbcnt_hi_u32_b32 %0, -1, %0" : "=v"(lane));
            return wave_base + lane;
        }

        // arrangement 1 -> memory: pairs, consecutive lanes 16 bytes apart
        template <int T>
        __device__ __forceinline__ void h_store_rows(const u64 (&x)[32], u64 *__restrict__ halfp, int tid)
        {
            const int jb = Arr<T, 1>::tid_index(tid);
#pragma unroll
            for (int s = 0; s < 32; s += 2)
                store_nt(halfp + jb + Arr<T, 1>::slot_index(s), x[s], x[s + 1]);
        }

        // reduce mode 7: the words of arrangement 1 are canonical residues t of temp_q (NTT form); what is stored is the rest of
        // the CKKS mod-down (NttSource::ModDownStore): v = (prod + t) * P^-1 mod q, into the ciphertext
        template <int T>
        __device__ __forceinline__ void h_store_moddown(const u64 (&x)[32], int tid, const u64 *__restrict__ prod_half,
                                                        u64 *__restrict__ ct_half, const u64 *__restrict__ c0_half, bool add_ct,
                                                        u64 inv_p, u64 inv_p_shoup, u64 p, unsigned *__restrict__ tflag)
        {
            const int jb = Arr<T, 1>::tid_index(tid);
            u64 nz = 0; // transparency sink: OR of the words stored into component 1 (tflag is null for component 0)
#pragma unroll
            for (int b = 0; b < 32; b += 8)
            {
                ulonglong2 pr[4], cc[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int off = jb + Arr<T, 1>::slot_index(b + 2 * i);
                    pr[i] = *reinterpret_cast<const ulonglong2 *>(prod_half + off);
                    if (c0_half)
                        cc[i] = *reinterpret_cast<const ulonglong2 *>(c0_half + off);
                    else if (add_ct)
                        cc[i] = *reinterpret_cast<const ulonglong2 *>(ct_half + off);
                }
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int off = jb + Arr<T, 1>::slot_index(b + 2 * i);
                    u64 v0 = mulmod_shoup(pr[i].x + x[b + 2 * i], inv_p, inv_p_shoup, p);
                    u64 v1 = mulmod_shoup(pr[i].y + x[b + 2 * i + 1], inv_p, inv_p_shoup, p);
                    if (c0_half || add_ct)
                    {
                        v0 = add_mod(v0, cc[i].x, p);
                        v1 = add_mod(v1, cc[i].y, p);
                    }
                    store_nt(ct_half + off, v0, v1);
                    nz |= v0 | v1;
                }
            }
            note_nonzero(tflag, 0, nz);
        }

        template <int LOGN, int STRICT, int REDUCE>
        __global__ __launch_bounds__(1 << (LOGN - 6), 4) void ntt_fwd_half_kernel(
            u64 *__restrict__ data, const PrimeDev *__restrict__ primes, RowMap map, std::size_t nrows, int flags,
            unsigned *__restrict__ tickets, unsigned *__restrict__ timeout_flag, unsigned spin_limit, NttSource src,
            std::size_t chunk, LiveSlots live)
        {
            constexpr int T = LOGN - 1;
            constexpr int N = 1 << LOGN;
            extern __shared__ u64 lds[];
            const int tid = threadIdx.x; // (only the measurement hooks below use it: see fresh_tid)
            const int wave_base = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) & ~63);
            int half, position;
            std::size_t poly;
            if (!half_block_map(blockIdx.x, nrows / map.rows, live.n, chunk, poly, position, half,
                                (flags & kNttPolyMajor) ? ((flags >> 16) & 0xFF) : 0))
                return;
            const std::size_t row = poly * map.rows + live.slot[position];
            const unsigned short pid = map.prime[row % map.rows];
            const PrimeDev P = primes[pid];
            constexpr bool FP = STRICT == 3; // butterflies on the FP64 pipe (primes below 2^50, see fp_reduce_all)
            const u64 p = P.p, two_p = FP ? fp_bits(P.p_d) : P.two_p, rdp = P.rdp;
            const u64 *tw = FP ? reinterpret_cast<const u64 *>(P.fwd_d) : P.fwd;
            u64 *rowp = data + (row << LOGN);
            u64 x[32];

            // ---- load both halves, top layer on the fly, arrangement 1 (block-uniform branch on the half)
            const u64 neg_p = FP ? fp_bits(P.pinv_d) : 0 - p;
            const u64 *srcp = rowp;
            if (src.base[0])
            {
                const unsigned short code = src.code[row % map.rows];
                if (code != kSkipRow)
                {
                    const int b = code >> 15;
                    srcp = src.base[b] + (row / map.rows) * src.poly_stride[b] +
                           (static_cast<std::size_t>(code & 0x3FFF) << LOGN);
                }
            }
            NTT_STAMP(0);
#ifdef SEALHIP_NTT_EXPERIMENT
            if ((flags & 0x2000) && tid == 0 && g_ntt_trace)
                g_ntt_trace[static_cast<std::size_t>(blockIdx.x) * 8 + 6] = __builtin_readcyclecounter();
#endif
            if (NTT_EXP(flags, 0x1000) && (blockIdx.x >> 3) >= 32 && (blockIdx.x >> 3) < 64)
            {
                for (int i = 0; i < ((flags >> 16) & 0xFF); i++)
                    __builtin_amdgcn_s_sleep(127);
            }
            if (NTT_EXP(flags, 0x400))
            {
#pragma unroll
                for (int i = 0; i < 32; i++)
                    x[i] = static_cast<u64>(tid) * 0x9E3779B97F4A7C15ull + i;
            }
            else if constexpr (REDUCE == 6)
            {
                // kNttTopDone: the producer applied the top layer; this workgroup's half, arrangement 1, nothing else
                const int jb1 = Arr<T, 1>::tid_index(fresh_tid(wave_base));
#pragma unroll
                for (int s = 0; s < 32; s += 2)
                {
                    const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(rowp + (half << T) + jb1 + Arr<T, 1>::slot_index(s));
                    x[s] = v.x;
                    x[s + 1] = v.y;
                }
            }
            else if constexpr (SEALHIP_NTT_XCHG_TOP && REDUCE <= 3)
            {
                {
                    // one flag word per (row, wave): bit h = "the wave of workgroup h has published"
                    unsigned *flagw = tickets + (row << 4) + (static_cast<unsigned>(wave_base) >> 6);
                    const bool signal = !(flags & kNttDebugNoSignal);
                    if (half)
                        h_load_top_xchg<T, STRICT, 1, REDUCE>(x, srcp, rowp, tw, fresh_tid(wave_base), two_p, neg_p, P.cr1, flagw,
                                                              timeout_flag, spin_limit, signal);
                    else
                        h_load_top_xchg<T, STRICT, 0, REDUCE>(x, srcp, rowp, tw, fresh_tid(wave_base), two_p, neg_p, P.cr1, flagw,
                                                              timeout_flag, spin_limit, signal);
                }
            }
            else if (half)
                h_load_top<T, STRICT, 1, REDUCE>(x, srcp, tw, fresh_tid(wave_base), two_p, neg_p, P.cr1, src.aux_p, src.aux_cr1, src.aux_top);
            else
                h_load_top<T, STRICT, 0, REDUCE>(x, srcp, tw, fresh_tid(wave_base), two_p, neg_p, P.cr1, src.aux_p, src.aux_cr1, src.aux_top);
            // The transform is in place and both workgroups of a row read BOTH halves: neither may store before
            // the other has finished loading. Ticket protocol (placement independent, bounded spin): every
            // wave bumps the row's counter once its loads have landed in registers; before its store phase
            // it waits until the counter shows all waves of both workgroups. Only a "finished reading" signal crosses workgroups, so
            // relaxed agent-scope atomics suffice (no payload is published).
            // The signal is sent after the first LDS exchange: its barriers are only passed once every wave of
            // the workgroup has consumed all of its loaded values in round 1, so no extra wait or barrier is needed.
            const int gbase = half << T;
            NTT_STAMP(1);
            // round 1: every lane index bit lies below the processed bits -> block-uniform twiddles
            u64 w0[kIL], ws0[kIL];
            RoundStage<T, 1, STRICT, true, 0>::load(w0, ws0, tw, gbase, N);
            if constexpr (FP)
                fp_reduce_all(x, two_p, neg_p);
            ZeroPairs zp; // (written again where each phase starts: two moves, and no register held across the exchanges)
            if constexpr (kApx<STRICT> == 2)
                zp.init();
            if (!NTT_EXP(flags, 0x100))
                RoundPipe<T, 1, STRICT, true>::run(x, w0, ws0, tw, gbase, N, two_p, neg_p, zp);
            const int jb2 = gbase + Arr<T, 2>::tid_index(fresh_tid(wave_base));
            RoundStage<T, 2, STRICT, false, 0>::load(w0, ws0, tw, jb2, N); // lands while the exchange runs
            __builtin_amdgcn_sched_barrier(0);
            if (!NTT_EXP(flags, 0x200))
                h_exchange<T, 1, 2>(x, lds, fresh_tid(wave_base));
            constexpr bool xchg_top = SEALHIP_NTT_XCHG_TOP && REDUCE <= 3; // (its own hand-off, in the load phase)
            if (!xchg_top && fresh_tid(wave_base) == 0 && tickets && !(flags & kNttDebugNoSignal))
                __hip_atomic_fetch_add(&tickets[row], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            if constexpr (kApx<STRICT> == 2)
                zp.init();
            if (!NTT_EXP(flags, 0x100))
                RoundPipe<T, 2, STRICT, false>::run(x, w0, ws0, tw, jb2, N, two_p, neg_p, zp);
            const int jb3 = gbase + Arr<T, 3>::tid_index(fresh_tid(wave_base));
            RoundStage<T, 3, STRICT, false, 0>::load(w0, ws0, tw, jb3, N);
            __builtin_amdgcn_sched_barrier(0);
            if (!NTT_EXP(flags, 0x200))
                h_exchange<T, 2, 3>(x, lds, fresh_tid(wave_base));
            if constexpr (kApx<STRICT> == 2)
                zp.init();
            if (!NTT_EXP(flags, 0x100))
                RoundPipe<T, 3, STRICT, false>::run(x, w0, ws0, tw, jb3, N, two_p, neg_p, zp);
            const int jb4 = gbase + Arr<T, 4>::tid_index(fresh_tid(wave_base));
            u64x2 tg0[FinalStage<T>::SG * FinalStage<T>::NTW];
            if constexpr (FinalStage<T>::PIPE)
            {
                StageTw<T, 0, 0, STRICT == 3>::load(tg0, tw, jb4, N); // lands while the last exchange runs
                __builtin_amdgcn_sched_barrier(0);
            }
            if (!NTT_EXP(flags, 0x200))
                h_exchange<T, 3, 4>(x, lds, fresh_tid(wave_base));
            static_assert(!bounds::fp_fwd_reduce_before_layer(12) && !bounds::fp_fwd_reduce_before_layer(13) &&
                              !bounds::fp_fwd_reduce_before_layer(14),
                          "the final round runs without a reduction (the schedule reduces inside rounds 2 and 3)");
            NTT_STAMP(2);
            // ---- wait until the sibling workgroup has read its inputs (normally true ~tens of microseconds ago)
            const auto wait_for_sibling = [&] {
                if (xchg_top)
                    return;
                if ((fresh_tid(wave_base) & 63) == 0 && tickets) // one poll per wave, no workgroup barrier
                {
                    unsigned spins = 0;
                    while (__hip_atomic_load(&tickets[row], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < 2u)
                    {
                        __builtin_amdgcn_s_sleep(8);
                        if (++spins > spin_limit)
                        {
                            // never observed outside the tests that force it; do not hang the device: flag the launch as
                            // failed (host-mapped word, read by every host-visible synchronisation point) and fall through
                            __hip_atomic_store(timeout_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_SYSTEM);
                            break;
                        }
                    }
                }
                // (the other lanes of the wave wait for lane 0 through re-convergence; every wave checks for itself
                //  that both workgroups of the row have finished reading)
            };
            constexpr bool XCH = kStoreExchange<T, STRICT, REDUCE>;
            constexpr int SX = XCH ? 1 : (kStoreSwap<T, STRICT, REDUCE> ? 2 : 0);
            if constexpr (!XCH)
                wait_for_sibling(); // the final round stores as it goes
            NTT_STAMP(3);
            // ---- final round + store, group by group (arrangement 4: runs of 2^f consecutive coefficients per lane)
            const int Nx = NTT_EXP(flags, 0xF00) ? (N | ((flags & 0xF00) << 20)) : N;
            // bit 0: canonicalising wrapper; bit 1: leave the last layer's first operand unreduced (kNttAnyRep)
            // bit 2 (approximate-quotient canonical launches on primes of at least 45 bits, launch_half): the canonicalising
            // step estimates its small quotient in single precision (devmath.hpp reduce_small_quot)
            // (STRICT == 4, the dense lazy schedule: the last layer leaves its first operand as it is, the store reduces)
            const int fin = ((flags & kNttCanonical) ? 1 : 0) | (((flags & kNttAnyRep) || STRICT == 4) ? 2 : 0) | ((flags & kNttSmallQuot) ? 4 : 0);
            const u64 rdp_fin =
                ((STRICT == 2 && (flags & kNttSmallQuot)) || STRICT == 4) ? static_cast<u64>(__float_as_uint(small_quot_const(p))) : rdp;
            constexpr bool ROUT = REDUCE == 3 || REDUCE == 6; // kNttReduceOut launches (never gathered: no load treatment to combine with)
            if constexpr (kFinalApx<T, STRICT> == 2)
                zp.init();
            if constexpr (FinalStage<T>::PIPE)
                FinalPipe<T, STRICT, ROUT, SX, 0>::run(x, tg0, tw, rowp, jb4, Nx, p, two_p, neg_p, rdp_fin, fin, zp);
            else
                FinalGroups<T, STRICT, 0, 1 << (5 - (T - 12)), ROUT, SX>::run(x, tw, rowp, jb4, Nx, p, two_p, neg_p, rdp_fin, fin, zp);
            if constexpr (XCH)
            {
                if (!NTT_EXP(flags, 0x200))
                    h_exchange<T, 4, 1>(x, lds, fresh_tid(wave_base));
                wait_for_sibling();
                if constexpr (REDUCE == 7)
                {
                    // row = (polynomial pl of the launch, prime slot q): products at prod[pl][q], ciphertext component pl & 1
                    const std::size_t pl = row / map.rows, q = row % map.rows;
                    typedef const __attribute__((address_space(4))) u64 *kc_t;
                    const u64 ip = ((kc_t)src.md.inv_p)[q], ips = ((kc_t)src.md.inv_p_shoup)[q];
                    const u64 *prod_half = src.md.prod + pl * src.md.prod_stride + (q << LOGN) + gbase;
                    u64 *ct_half = src.md.ct + (pl >> 1) * src.md.ct_stride + (((pl & 1) * map.rows + q) << LOGN) + gbase;
                    const u64 *c0_half =
                        (src.md.c0_src && !(pl & 1)) ? src.md.c0_src + (pl >> 1) * src.md.c0_stride + (q << LOGN) + gbase : nullptr;
                    h_store_moddown<T>(x, fresh_tid(wave_base), prod_half, ct_half, c0_half, src.md.c0_src == nullptr, ip, ips, P.p,
                                       (src.md.tflags && (pl & 1)) ? src.md.tflags + (pl >> 1) : nullptr);
                }
                else
                    h_store_rows<T>(x, rowp + gbase, fresh_tid(wave_base));
            }
            NTT_STAMP(4);
#ifdef SEALHIP_NTT_EXPERIMENT
            if ((flags & 0x2000) && tid == 0 && g_ntt_trace)
            {
                unsigned hw;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
                unsigned xcc;
                asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
                g_ntt_trace[static_cast<std::size_t>(blockIdx.x) * 8 + 5] = (static_cast<u64>(xcc) << 32) | hw;
                g_ntt_trace[static_cast<std::size_t>(blockIdx.x) * 8 + 7] = __builtin_readcyclecounter();
            }
#endif
        }

        // ----------------------------------------------------------------------------------------
        // Single-pass inverse NTT for logn = 14..16: the mirror image of ntt_fwd_half_kernel. The
        // Gentleman-Sande layers on index bits 0..T-1 only pair coefficients inside one half of the row, so
        // a workgroup transforms its half entirely on chip (final arrangement first, then rounds 3, 2, 1,
        // ascending bits) and stores lazy values in [0, 2p). The top layer (gap N/2, with n^{-1} folded
        // in, ntt.cpp:393-402) needs both halves and is applied by ntt_inv_top_kernel, a pure streaming
        // pass (or, inside the pipelines, by the consumer kernel).
        // ---- inverse rounds as stage pipelines (mirror of RoundStage / RoundPipe; layers ascend W = 1, 2, 3, 4)
        // Lazy-sum schedule of the inverse (only when the caller accepts any representative of the stored values and every
        // prime of the launch is small enough, launch_half_inv): the conditional subtraction of the sum output is dropped
        // on all but two of the T on-chip layers; those two (the middle one and the last) reduce with barrett_lazy
        // instead. Values entering layer l are below 2^shift(l) * p; the difference operand gets that bound added.
        // (the schedule, its worst-case recurrence and the admission predicate live in ntt_bounds.hpp)
        // Round 4: the MODE 1 layers of the lazy schedule (all but two) take the level-2 quotient (devmath.hpp mulhi_apx2,
        // butterflies_inv_apx2): their products land below 4p, which is what the unreduced sum of such a layer is bounded by
        // anyway, so shift(), mode() and the admission predicate are what they were (ntt_bounds.hpp section 1).
#ifndef SEALHIP_NTT_INV_APX2
#define SEALHIP_NTT_INV_APX2 1
#endif
        constexpr bool kInvApx2 = SEALHIP_NTT_INV_APX2 != 0;
        // LZ 1: the sparse schedule (two reducing layers; primes with head-room), LZ 3 (round 4): the DENSE schedule for primes
        // up to 2^60 -- every third layer and the last reduce their sums with the single-precision quotient estimate, values
        // never pass 16p (ntt_bounds.hpp section 1): the 60-bit Bsk rows of a BFV multiply and ciphertext primes of 56-60 bits
        // no longer pay a conditional subtraction in every butterfly.
        template <int LZ>
        constexpr bool kLazy = LZ == 1 || LZ == 3;
        template <int T, int LZ = 1>
        struct InvLazy
        {
            static constexpr int sched = LZ == 3 ? 1 : 0;
            static constexpr int mode(int l)
            {
                return bounds::inv_lazy_mode(T, l, sched);
            }
            static constexpr int shift(int l)
            {
                return bounds::inv_lazy_shift(T, l, sched);
            }
            // what butterflies_inv_hs gets for a reducing layer: Barrett (sparse) or the quotient estimate (dense)
            static constexpr int reduce_mode = LZ == 3 ? 3 : 2;
        };
        // on-chip layers of the inverse kernel instance ntt_inv_half_kernel<KLOGN, ...> (half-row form of a ring of 2^KLOGN,
        // or whole-row form of a ring of 2^(KLOGN-1)): what the launchers hand to bounds::inv_lazy_admits
        template <int KLOGN>
        constexpr int kInvLayers = KLOGN - 1;
        // Floating-point schedule of the inverse (LZ == 2, primes below 2^50, inputs below 2p): sums double the bound per
        // layer, so both outputs of layers 1, 5, 9, 13 are brought back to [-p/2, p/2]: 2p -> 4p -> 8p | 0.5p -> p -> 2p ->
        // 4p -> 8p | ...; a difference is at most 8p too, its product below (0.5 + 8p 2^-52) p <= 2.5p. Every magnitude stays
        // at or below 8p < 2^53: exact (devmath.hpp). The last layer's outputs are canonicalised by the store instead.
        // p << shift as a value of its own at every use. The subtraction u - y + addend is a 64-bit v_sub / v_subb pair; the
        // second reads the carry, so its other operand cannot be a scalar register and the compiler keeps the addend's high
        // dword in a VECTOR register -- and, because the same shift recurs in layers far apart, kept it there (or in scratch:
        // three spilled dwords in <16, 1, true>) across whole rounds. An opaque scalar copy per layer ends that live range.
        __device__ __forceinline__ u64 lazy_addend(u64 neg_p, int shift)
        {
            u64 a = (0 - neg_p) << shift;
            asm volatile("" : "+s"(a));
            return a;
        }
        template <int T>
        constexpr bool fp_inv_reduce_after(int layer)
        {
            return bounds::fp_inv_reduce_after_layer(T, layer);
        }
        template <int T, int R, bool UNIFORM, int K, int LZ = 0>
        struct RoundStageInv
        {
            static constexpr int PER = 16 / kIL;
            static constexpr int W = 1 + K / PER;
            static constexpr int C = (K % PER) * kIL;
            static constexpr int bit = 1 << W;
            static constexpr int slot(int j)
            {
                return (((C + j) >> W) << (W + 1)) | ((C + j) & (bit - 1));
            }
            __device__ static __forceinline__ void load(u64 (&w)[kIL], u64 (&ws)[kIL], const u64 *__restrict__ tw, int jb,
                                                        int N)
            {
                const int tb = (N + jb) >> (Arr<T, R>::slot_bit(W) + 1);
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    u64x2 Wv;
                    if constexpr (LZ == 2)
                    {
                        Wv.x = UNIFORM ? ((twd_const_t)tw)[__builtin_amdgcn_readfirstlane(tb) + Arr<T, R>::tw_offset(slot(j), W)]
                                       : ((twd_global_t)tw)[tb + Arr<T, R>::tw_offset(slot(j), W)];
                        Wv.y = 0;
                    }
                    else if (UNIFORM)
                        Wv = ((tw_const_t)tw)[__builtin_amdgcn_readfirstlane(tb) + Arr<T, R>::tw_offset(slot(j), W)];
                    else
                        Wv = ((tw_global_t)tw)[tb + Arr<T, R>::tw_offset(slot(j), W)];
                    w[j] = Wv.x;
                    ws[j] = Wv.y;
                }
            }
            static constexpr int layer = (T - 12) + 4 * (3 - R) + (W - 1); // 0-based on-chip layer index
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64 (&w)[kIL], const u64 (&ws)[kIL], u64 two_p,
                                                       u64 neg_p, u64 rdp, ZeroPairs &zp)
            {
                if constexpr (LZ == 2)
                {
                    const double pd = fp_of(two_p), pinv = fp_of(neg_p);
#pragma unroll
                    for (int j = 0; j < kIL; j++)
                    {
                        fp_butterfly_inv(x[slot(j)], x[slot(j) | bit], w[j], pd, pinv);
                        if constexpr (fp_inv_reduce_after<T>(layer))
                        {
                            x[slot(j)] = fp_bits(fp_reduce(fp_of(x[slot(j)]), pd, pinv));
                            x[slot(j) | bit] = fp_bits(fp_reduce(fp_of(x[slot(j) | bit]), pd, pinv));
                        }
                    }
                    return;
                }
                u64 u[kIL], y[kIL];
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    u[j] = x[slot(j)];
                    y[j] = x[slot(j) | bit];
                }
                if constexpr (kLazy<LZ> && InvLazy<T, LZ>::mode(layer) == 1 && kInvApx2)
                    butterflies_inv_apx2<UNIFORM, kIL>(u, y, w, ws, neg_p, lazy_addend(neg_p, InvLazy<T, LZ>::shift(layer)), zp.z);
                else if constexpr (kLazy<LZ>)
                    butterflies_inv_hs<UNIFORM, kIL, InvLazy<T, LZ>::mode(layer) == 1 ? 1 : InvLazy<T, LZ>::reduce_mode>(
                        u, y, w, ws, neg_p, lazy_addend(neg_p, InvLazy<T, LZ>::shift(layer)), rdp);
                else
                    butterflies_inv_hs<UNIFORM, kIL>(u, y, w, ws, neg_p, two_p); // BackwardLazy, ntt.cpp:265-272
#pragma unroll
                for (int j = 0; j < kIL; j++)
                {
                    x[slot(j)] = u[j];
                    x[slot(j) | bit] = y[j];
                }
            }
        };
        template <int T, int R, bool UNIFORM, int LZ, int K = 0, int NLAYERS = 4>
        struct RoundPipeInv
        {
            static constexpr int NST = NLAYERS * (16 / kIL); // (NLAYERS = 3: the whole-row form applies the round's last layer itself)
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64 (&w)[kIL], const u64 (&ws)[kIL],
                                                       const u64 *__restrict__ tw, int jb, int N, u64 two_p, u64 neg_p,
                                                       u64 rdp, ZeroPairs &zp)
            {
                u64 wn[kIL], wsn[kIL];
                if constexpr (K + 1 < NST)
                    RoundStageInv<T, R, UNIFORM, K + 1, LZ>::load(wn, wsn, tw, jb, N);
                __builtin_amdgcn_sched_barrier(0);
                RoundStageInv<T, R, UNIFORM, K, LZ>::run(x, w, ws, two_p, neg_p, rdp, zp);
                if constexpr (K + 1 < NST)
                    RoundPipeInv<T, R, UNIFORM, LZ, K + 1, NLAYERS>::run(x, wn, wsn, tw, jb, N, two_p, neg_p, rdp, zp);
            }
        };

        // first phase of the inverse: the 2^f consecutive coefficients that share the filler slot bits G run their low
        // layers (index bits 0 .. f-1, ascending); group twiddles in the order used: layer W = 0 (2^(f-1) entries),
        // W = 1, ..., W = f-1 (1 entry). All coefficients are loaded before (one exposed latency), twiddles are
        // requested one stage (FinalStage<T>::SG groups) ahead.
        template <int T, int G, bool FP = false>
        __device__ __forceinline__ void h_first_tw(u64x2 *tg, const u64 *__restrict__ tw, int jb, int N)
        {
            constexpr int f = T - 12;
            int base = 0;
#pragma unroll
            for (int W = 0; W < f; W++)
            {
                const int tb = (N + jb) >> (Arr<T, 4>::slot_bit(W) + 1);
#pragma unroll
                for (int o = 0; o < (1 << (f - 1 - W)); o++)
                {
                    const int s = (G << f) | (o << (W + 1));
                    if constexpr (FP)
                        tg[base + o].x = ((twd_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)];
                    else
                        tg[base + o] = ((tw_global_t)tw)[tb + Arr<T, 4>::tw_offset(s, W)];
                }
                base += 1 << (f - 1 - W);
            }
        }
#ifndef SEALHIP_NTT_INV_FIRST_APX2
#define SEALHIP_NTT_INV_FIRST_APX2 1 // (level-2 quotient in the non-reducing layers of the first round too; 0: exact, as before)
#endif
        template <int T, int G, int LZ>
        __device__ __forceinline__ void h_first_group_regs(u64 (&x)[32], const u64x2 *tg, u64 neg_p, u64 two_p, u64 rdp, ZeroPairs &zp)
        {
            constexpr int f = T - 12;
            static_assert(LZ != 1 || f - 1 < bounds::inv_lazy_r1(T), "sparse schedule: the first layers are never the reducing ones");
            int base = 0;
#pragma unroll
            for (int W = 0; W < f; W++)
            {
                const int bit = 1 << W;
                const u64 addend = kLazy<LZ> ? lazy_addend(neg_p, InvLazy<T, LZ>::shift(W)) : two_p; // layer index = W
#pragma unroll
                for (int e = 0; e < (1 << f); e++)
                {
                    if (e & bit)
                        continue;
                    const int s = (G << f) | e;
                    const u64x2 Wv = tg[base + (e >> (W + 1))];
                    if constexpr (LZ == 2)
                    {
                        const double pd = fp_of(two_p), pinv = fp_of(neg_p);
                        fp_butterfly_inv(x[s], x[s | bit], Wv.x, pd, pinv);
                        if (fp_inv_reduce_after<T>(W))
                        {
                            x[s] = fp_bits(fp_reduce(fp_of(x[s]), pd, pinv));
                            x[s | bit] = fp_bits(fp_reduce(fp_of(x[s | bit]), pd, pinv));
                        }
                        continue;
                    }
                    const u64 u = x[s], v = x[s | bit];
                    u64 tt = u + v;
                    if (!kLazy<LZ>)
                        tt = tt >= two_p ? tt - two_p : tt;
                    else if (InvLazy<T, LZ>::mode(W) != 1) // (dense schedule at N = 2^16: its layer 2 is one of the first three)
                        tt = reduce_small_quot(tt, __uint_as_float(static_cast<unsigned>(rdp)), neg_p);
                    x[s] = tt;
                    // (W is a constant after unrolling: the branch folds)
                    // (sparse schedule only: the dense plain instances spill four dwords with the pairs live here)
                    if (SEALHIP_NTT_INV_FIRST_APX2 && LZ == 1 && kInvApx2 && InvLazy<T, 1>::mode(W) == 1)
                        x[s | bit] = mulmod_lazy_apx2<false>(u - v + addend, Wv.x, Wv.y, neg_p, zp.z[(e >> (W + 1)) & 1]);
                    else
                        x[s | bit] = mulmod_lazy_hs<false>(u - v + addend, Wv.x, Wv.y, neg_p);
                }
                base += 1 << (f - 1 - W);
            }
        }
        template <int T, int ST, int LZ, int I = 0>
        struct FirstStage
        {
            __device__ static __forceinline__ void load(u64x2 *tg, const u64 *__restrict__ tw, int jb, int N)
            {
                h_first_tw<T, ST * FinalStage<T>::SG + I, LZ == 2>(tg + I * FinalStage<T>::NTW, tw, jb, N);
                if constexpr (I + 1 < FinalStage<T>::SG)
                    FirstStage<T, ST, LZ, I + 1>::load(tg, tw, jb, N);
            }
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64x2 *tg, u64 neg_p, u64 two_p, u64 rdp, ZeroPairs &zp)
            {
                h_first_group_regs<T, ST * FinalStage<T>::SG + I, LZ>(x, tg + I * FinalStage<T>::NTW, neg_p, two_p, rdp, zp);
                if constexpr (I + 1 < FinalStage<T>::SG)
                    FirstStage<T, ST, LZ, I + 1>::run(x, tg, neg_p, two_p, rdp, zp);
            }
        };
        // AHEAD: the next stage's twiddles are requested before this stage is computed (two stages of twiddles live: 48
        // registers at f = 2). Off at f = 3 (they do not fit), in the lazy fused-tensor instances, which come out of their
        // products at the register cap, and in the exact whole-row instance (the request then follows the stage: its latency is
        // exposed once per stage; round 3: with this, fresh_tid and lazy_addend no single-pass kernel spills any more).
        template <int T, int ST, int LZ, bool AHEAD = FinalStage<T>::PIPE>
        struct FirstPipe
        {
            __device__ static __forceinline__ void run(u64 (&x)[32], const u64x2 *cur, const u64 *__restrict__ tw, int jb,
                                                       int N, u64 neg_p, u64 two_p, u64 rdp, ZeroPairs &zp)
            {
                u64x2 next[FinalStage<T>::SG * FinalStage<T>::NTW];
                if constexpr (ST + 1 < FinalStage<T>::NS && AHEAD)
                    FirstStage<T, ST + 1, LZ>::load(next, tw, jb, N);
                __builtin_amdgcn_sched_barrier(0);
                FirstStage<T, ST, LZ>::run(x, cur, neg_p, two_p, rdp, zp);
                __builtin_amdgcn_sched_barrier(0);
                if constexpr (ST + 1 < FinalStage<T>::NS && !AHEAD)
                    FirstStage<T, ST + 1, LZ>::load(next, tw, jb, N);
                if constexpr (ST + 1 < FinalStage<T>::NS)
                    FirstPipe<T, ST + 1, LZ, AHEAD>::run(x, next, tw, jb, N, neg_p, two_p, rdp, zp);
            }
        };

        // ---- ciphertext tensor product formed on load (evaluator.cpp:376-420 for two size-2 operands): output polynomial I
        // of an item is c_0 = a_0 b_0, c_1 = a_0 b_1 + a_1 b_0, c_2 = a_1 b_1 over the forward-transformed rows X[s] (s = 0, 1:
        // a; s = 2, 3: b) of the same prime. The reference reduces every product with Barrett and adds with one conditional
        // subtraction; its results are canonical residues that only feed this inverse transform, so any representative of
        // the same residue class below 2p gives the same final output. Here: carry-free 128-bit sum of products (operands
        // below 2^61) and ONE Montgomery reduction, which leaves the factor 2^-64; the consumer's constants carry 2^64
        // (RnsDev::floor_*_topM). The operands must make that reduction land below 2p: below 4p for primes under 2^59
        // (what the lazy forward transform stores), below 2p for primes up to 2^61 -- the wrapped 60-bit Bsk rows hold
        // arbitrary 64-bit words (which dyadic_product_coeffmod accepts, polyarithsmallmod.cpp:63-117), so the forward
        // launch that produces them reduces every word with barrett_lazy before it stores it (kNttReduceOut).
#ifndef SEALHIP_NTT_TENSOR_GROUPED
#define SEALHIP_NTT_TENSOR_GROUPED 1
#endif
        struct DyadicSrc
        {
            const u64 *x;                        // forward-transformed operands: item-major, 4 polynomials of kb rows
            std::size_t item_stride, poly_stride; // words
            int kb;
            // Evaluator::square (evaluator.cpp:560-702): TWO polynomials per item; c_0 = x_0^2, c_1 = x_0 x_1 added to itself
            // (:650-651), c_2 = x_1^2
            int square;
        };
        // IL words in lock step: t_j = (sum of NP products of operands below 2^61) * 2^-64 mod p as a Montgomery reduction,
        // t_j < sum / 2^64 + p. 4 multiplier instructions per product (the operands' upper halves are below 2^29, so the
        // middle sums cannot overflow), 3 for m = lo * (-p^-1) mod 2^64, 4 + one carry for floor(m p / 2^64).
        // Program-ordered (volatile) like the butterflies: consecutive instructions belong to different words, and the
        // carry of the high product is read IL >= 3 instructions after it is written.
        template <int IL, int NP>
        __device__ __forceinline__ void dyadic_redc(u64 (&t)[IL], const u64 (&a)[NP][IL], const u64 (&b)[NP][IL], u64 p, u64 ninv)
        {
            static_assert(IL >= 3, "the carry of the high product is read IL instructions after its producer");
            typedef unsigned __int128 u128;
            u64 P0[NP][IL], M[IL], H[IL], cy[IL] = {};
#pragma unroll
            for (int q = 0; q < NP; q++)
            {
#pragma unroll
                for (int j = 0; j < IL; j++)
                    P0[q][j] = mul64v<false>(static_cast<u32>(a[q][j]), static_cast<u32>(b[q][j]), cy[j]);
#pragma unroll
                for (int j = 0; j < IL; j++)
                    M[j] = q == 0 ? mul64v<false>(static_cast<u32>(a[q][j]), static_cast<u32>(b[q][j] >> 32), cy[j])
                                  : mad64v<false>(static_cast<u32>(a[q][j]), static_cast<u32>(b[q][j] >> 32), M[j], cy[j]);
#pragma unroll
                for (int j = 0; j < IL; j++)
                    M[j] = mad64v<false>(static_cast<u32>(a[q][j] >> 32), static_cast<u32>(b[q][j]), M[j], cy[j]);
#pragma unroll
                for (int j = 0; j < IL; j++)
                    H[j] = q == 0 ? mul64v<false>(static_cast<u32>(a[q][j] >> 32), static_cast<u32>(b[q][j] >> 32), cy[j])
                                  : mad64v<false>(static_cast<u32>(a[q][j] >> 32), static_cast<u32>(b[q][j] >> 32), H[j], cy[j]);
            }
            u64 lo[IL], hi[IL], m[IL], A[IL], B[IL], carry[IL], mh[IL];
            u32 cb[IL];
            const u32 p0 = static_cast<u32>(p), p1 = static_cast<u32>(p >> 32);
#pragma unroll
            for (int j = 0; j < IL; j++)
            {
                u128 X = (static_cast<u128>(H[j]) << 64) + (static_cast<u128>(M[j]) << 32);
#pragma unroll
                for (int q = 0; q < NP; q++)
                    X += P0[q][j];
                lo[j] = static_cast<u64>(X);
                hi[j] = static_cast<u64>(X >> 64);
                m[j] = lo[j] * ninv;
            }
#pragma unroll
            for (int j = 0; j < IL; j++)
                A[j] = mad64v<true>(static_cast<u32>(m[j] >> 32), p0, static_cast<u64>(__umulhi(static_cast<u32>(m[j]), p0)), cy[j]);
#pragma unroll
            for (int j = 0; j < IL; j++)
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(B[j]), "=s"(carry[j]) : "v"(static_cast<u32>(m[j])), "s"(p1), "v"(A[j]));
#pragma unroll
            for (int j = 0; j < IL; j++)
                asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(cb[j]) : "s"(carry[j]));
#pragma unroll
            for (int j = 0; j < IL; j++)
                mh[j] = mad64v<true>(static_cast<u32>(m[j] >> 32), p1,
                                     static_cast<u64>(static_cast<u32>(B[j] >> 32)) | (static_cast<u64>(cb[j]) << 32), cy[j]);
#pragma unroll
            for (int j = 0; j < IL; j++)
                t[j] = hi[j] + mh[j] + (lo[j] != 0); // lo + m p is a multiple of 2^64: its low word carries iff lo != 0
        }
        // The inverse starts in arrangement 4 (runs of 2^f consecutive coefficients per lane). Loading the rows that way
        // means 16-byte pieces at a 2^f * 8-byte stride per instruction for f >= 2 -- the mirror image of the store pattern
        // described at kStoreExchange. With SEALHIP_NTT_LOAD_EXCHANGE=1 the rows are loaded in arrangement 1 for f >= 2
        // (contiguous kilobytes per instruction; the tensor products are position-independent and are formed there) and
        // moved to arrangement 4 through the LDS. Measured (profiles/r02/ntt_store_pattern.txt): no gain -- unlike the
        // stores, the second load instruction of a line hits in the CU's L1 (plain inverse 3.85 vs 4.01 M rows/s at
        // N = 2^15, tensor inverse 14.1 vs 14.0 ms per 1024 pairs) -- so it stays off.
#ifndef SEALHIP_NTT_LOAD_EXCHANGE
#define SEALHIP_NTT_LOAD_EXCHANGE 0
#endif
        template <int T>
        constexpr int kInvLoadArr = (SEALHIP_NTT_LOAD_EXCHANGE != 0 && (T - 12) >= 2) ? 1 : 4;

        // the half row's 32 words per lane, arrangement A, as products of two (c_0, c_2) or four (c_1) input rows
        // SAME: b is a (a square: one load). TWICE: the product added to itself, t + t below 4p brought back below 2p with
        // one conditional subtraction (the reference's add_poly_coeffmod of the product with itself: same residue).
        template <int T, int A, bool SAME = false, bool TWICE = false>
        __device__ __forceinline__ void h_load_dyadic2(u64 (&x)[32], const u64 *__restrict__ a, const u64 *__restrict__ b,
                                                       int jloc, u64 p, u64 ninv)
        {
#pragma unroll
            for (int batch = 0; batch < 4; batch++)
            {
                ulonglong2 va[4], vb[4];
#pragma unroll
                for (int i = 0; i < 4; i++)
                {
                    const int idx = jloc + Arr<T, A>::slot_index((batch * 4 + i) * 2);
                    va[i] = *reinterpret_cast<const ulonglong2 *>(a + idx);
                    if constexpr (SAME)
                        vb[i] = va[i];
                    else
                        vb[i] = *reinterpret_cast<const ulonglong2 *>(b + idx);
                }
#pragma unroll
                for (int g = 0; g < 2; g++)
                {
                    const u64 aa[1][4] = { { va[2 * g].x, va[2 * g].y, va[2 * g + 1].x, va[2 * g + 1].y } };
                    const u64 bb[1][4] = { { vb[2 * g].x, vb[2 * g].y, vb[2 * g + 1].x, vb[2 * g + 1].y } };
                    u64 t[4];
                    dyadic_redc<4, 1>(t, aa, bb, p, ninv);
#pragma unroll
                    for (int j = 0; j < 4; j++)
                    {
                        if constexpr (TWICE)
                        {
                            const u64 d = t[j] << 1, two_p = p << 1; // t below 2p <= 2^62
                            t[j] = d >= two_p ? d - two_p : d;
                        }
                        x[(batch * 4 + 2 * g) * 2 + j] = t[j];
                    }
                }
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        template <int T, int A>
        __device__ __forceinline__ void h_load_dyadic4(u64 (&x)[32], const u64 *__restrict__ a0, const u64 *__restrict__ b1,
                                                       const u64 *__restrict__ a1, const u64 *__restrict__ b0, int jloc, u64 p,
                                                       u64 ninv)
        {
#pragma unroll
            for (int batch = 0; batch < 8; batch++)
            {
                ulonglong2 v0[2], v1[2], v2[2], v3[2];
#pragma unroll
                for (int i = 0; i < 2; i++)
                {
                    const int idx = jloc + Arr<T, A>::slot_index((batch * 2 + i) * 2);
                    v0[i] = *reinterpret_cast<const ulonglong2 *>(a0 + idx);
                    v1[i] = *reinterpret_cast<const ulonglong2 *>(b1 + idx);
                    v2[i] = *reinterpret_cast<const ulonglong2 *>(a1 + idx);
                    v3[i] = *reinterpret_cast<const ulonglong2 *>(b0 + idx);
                }
                const u64 aa[2][4] = { { v0[0].x, v0[0].y, v0[1].x, v0[1].y }, { v2[0].x, v2[0].y, v2[1].x, v2[1].y } };
                const u64 bb[2][4] = { { v1[0].x, v1[0].y, v1[1].x, v1[1].y }, { v3[0].x, v3[0].y, v3[1].x, v3[1].y } };
                u64 t[4];
                dyadic_redc<4, 2>(t, aa, bb, p, ninv);
#pragma unroll
                for (int j = 0; j < 4; j++)
                    x[batch * 4 + j] = t[j];
                __builtin_amdgcn_sched_barrier(0);
            }
        }

        // WHOLE: the same workgroup shape (2^T coefficients, T = LOGN - 1) applied to a whole row of a ring of 2^T coefficients:
        // one workgroup per row, all T layers on chip -- the last of them is the row's top layer (BackwardLazyLast with
        // n^-1 folded in, ntt.cpp:274-281) -- so the standalone inverse is ONE launch that reads and writes the row once,
        // instead of the half-row kernel plus the streaming top-layer pass. All three arithmetic forms, N = 2^14 and 2^15.
        // QUARTER (round 4): the same shape applied to a QUARTER of a row of a ring of 2^(T + 2) coefficients -- for N = 2^16, where a
        // half row is 1024 lanes and the whole register file of a CU (one workgroup per CU, phases that cannot overlap: 0.38 FP64 /
        // 0.30 integer of the roofline), the N = 2^15 shape on quarter rows keeps two workgroups per CU (0.51 / 0.40 on the same
        // bytes, profiles/r04/n65536_quarter_row_projection.txt). It finishes index bits 0 .. T - 1 = 0 .. 13; the two layers
        // above (gap N/4 and the top layer, n^-1 folded in) are ntt_inv_top2_kernel's, one streaming radix-4 pass.
        template <int LOGN, int LZ, bool DY, bool WHOLE = false, bool QUARTER = false>
        __global__ __launch_bounds__(1 << (LOGN - 6), 4) void ntt_inv_half_kernel(u64 *__restrict__ data,
                                                                                  const PrimeDev *__restrict__ primes,
                                                                                  RowMap map, std::size_t nrows,
                                                                                  std::size_t chunk,
                                                                                  const u64 *__restrict__ src,
                                                                                  std::size_t src_poly_stride,
                                                                                  LiveSlots live, DyadicSrc dy,
                                                                                  int canonical = 0)
        {
            static_assert(!WHOLE || !DY, "whole-row form: plain transforms");
            static_assert(!QUARTER || (!DY && !WHOLE), "quarter-row form: plain transforms");
            constexpr int T = LOGN - 1;
            constexpr int LOGR = WHOLE ? T : (QUARTER ? T + 2 : LOGN); // log2 of the row length
            constexpr int N = 1 << LOGR;
            extern __shared__ u64 lds[];
            const int wave_base = __builtin_amdgcn_readfirstlane(static_cast<int>(threadIdx.x) & ~63); // (see fresh_tid)
            int half = 0, position;
            std::size_t poly;
            if constexpr (WHOLE)
            {
                // same XCD-aware enumeration as half_block_map, one workgroup per live row
                const std::size_t slot = blockIdx.x >> 3, npolys = nrows / map.rows;
                const std::size_t v = static_cast<std::size_t>(blockIdx.x & 7u) * chunk + slot;
                if (slot >= chunk || v >= npolys * static_cast<std::size_t>(live.n))
                    return;
                poly = v % npolys;
                position = static_cast<int>(v / npolys);
            }
            else if constexpr (DY && SEALHIP_NTT_TENSOR_GROUPED)
            {
                // The three output polynomials of one (item, prime) read the same four input rows (c_0: a_0 b_0, c_1: all
                // four, c_2: a_1 b_1): enumerate them next to each other -- v = (prime position, item, output) -- so that the six
                // workgroups run on one XCD at the same time and every input half row is fetched from HBM once and found in
                // that XCD's L2 by its second reader (prime-major as before: the twiddle tables stay L2-resident).
                // live.slot[] is sorted by slot = I * kb + r, i.e. three runs of the same nr prime rows.
                const unsigned xcd = blockIdx.x & 7u;
                const std::size_t slot = blockIdx.x >> 3, npolys = nrows / map.rows;
                half = static_cast<int>(slot & 1);
                const std::size_t v = static_cast<std::size_t>(xcd) * chunk + (slot >> 1);
                if ((slot >> 1) >= chunk || v >= npolys * static_cast<std::size_t>(live.n))
                    return;
                const int nr = live.n / 3;
                const std::size_t pr = v / 3;
                position = static_cast<int>(v - pr * 3) * nr + static_cast<int>(pr / npolys);
                poly = pr % npolys;
            }
            else if constexpr (QUARTER)
            {
                // four workgroups per live row, next to each other on one XCD; `half` counts quarters here (gbase = half << T)
                const std::size_t slot = blockIdx.x >> 3, npolys = nrows / map.rows;
                half = static_cast<int>(slot & 3);
                const std::size_t v = static_cast<std::size_t>(blockIdx.x & 7u) * chunk + (slot >> 2);
                if ((slot >> 2) >= chunk || v >= npolys * static_cast<std::size_t>(live.n))
                    return;
                poly = v % npolys;
                position = static_cast<int>(v / npolys);
            }
            else if (!half_block_map(blockIdx.x, nrows / map.rows, live.n, chunk, poly, position, half))
                return;
            const std::size_t row = poly * map.rows + live.slot[position];
            const unsigned short pid = map.prime[row % map.rows];
            const PrimeDev P = primes[pid];
            constexpr bool FP = LZ == 2; // two_p / neg_p then carry the bits of p and 1/p as doubles (see fp_reduce_all)
            static_assert(!FP || !DY, "the floating-point instance serves plain half transforms only");
            const u64 p = P.p, two_p = FP ? fp_bits(P.p_d) : P.two_p;
            const u64 *tw = FP ? reinterpret_cast<const u64 *>(P.inv_d) : P.inv;
            const int gbase = half << T;
            u64 *halfp = data + (row << LOGR) + gbase;
            // optional out-of-place input (polynomial-strided rows of another buffer): saves a copy kernel
            const u64 *inp = src ? src + (row / map.rows) * src_poly_stride + ((row % map.rows) << LOGR) + gbase : halfp;
            u64 x[32];
            const u64 neg_p = FP ? fp_bits(P.pinv_d) : 0 - p;
            // what the reducing layers of the lazy schedules read: floor(2^64 / p) (sparse: Barrett) or the bits of the
            // single-precision quotient constant (dense)
            const u64 rdp = LZ == 1 ? P.rdp : (LZ == 3 ? static_cast<u64>(__float_as_uint(small_quot_const(p))) : 0);
            {
                // every coefficient of the half row first (16 x 16 bytes per lane in flight at once), the twiddles of
                // the first stage with them
                constexpr int LA = kInvLoadArr<T>;
                const int jloc = Arr<T, 4>::tid_index(fresh_tid(wave_base)), jl = Arr<T, LA>::tid_index(fresh_tid(wave_base));
                u64x2 tg0[FinalStage<T>::SG * FinalStage<T>::NTW];
                if constexpr (!DY && LA == 4)
                    FirstStage<T, 0, LZ>::load(tg0, tw, gbase + jloc, N);
                if constexpr (DY)
                {
                    // map.rows = 3 * kb: slot = I * kb + r selects the output polynomial I and the prime row r
                    const int slot = live.slot[position], I = slot / dy.kb, r = slot - I * dy.kb;
                    const u64 *xr = dy.x + poly * dy.item_stride + (static_cast<std::size_t>(r) << LOGN) + gbase;
                    const std::size_t ps = dy.poly_stride;
                    if (dy.square) // (launch-uniform) two polynomials per item
                    {
                        if (I == 1)
                            h_load_dyadic2<T, LA, false, true>(x, xr, xr + ps, jl, p, P.ninv);
                        else
                            h_load_dyadic2<T, LA, true>(x, xr + (I >> 1) * ps, nullptr, jl, p, P.ninv);
                    }
                    else if (I == 1) // block-uniform
                        h_load_dyadic4<T, LA>(x, xr, xr + 3 * ps, xr + ps, xr + 2 * ps, jl, p, P.ninv);
                    else if (I == 0)
                        h_load_dyadic2<T, LA>(x, xr, xr + 2 * ps, jl, p, P.ninv);
                    else
                        h_load_dyadic2<T, LA>(x, xr + ps, xr + 3 * ps, jl, p, P.ninv);
                    // (the first stage's twiddles only now: held across the products they cost 24 registers of scratch)
                    if constexpr (LA == 4)
                        FirstStage<T, 0, LZ>::load(tg0, tw, gbase + jloc, N);
                }
                else
                {
#pragma unroll
                    for (int s = 0; s < 32; s += 2)
                    {
                        const ulonglong2 v = *reinterpret_cast<const ulonglong2 *>(inp + jl + Arr<T, LA>::slot_index(s));
                        x[s] = FP ? fp_bits(fp_from_u64(v.x)) : v.x; // (inputs below 2p < 2^52)
                        x[s + 1] = FP ? fp_bits(fp_from_u64(v.y)) : v.y;
                    }
                }
                if constexpr (LA != 4)
                {
                    FirstStage<T, 0, LZ>::load(tg0, tw, gbase + jloc, N); // lands while the exchange runs
                    __builtin_amdgcn_sched_barrier(0);
                    h_exchange<T, LA, 4>(x, lds, fresh_tid(wave_base));
                }
                __builtin_amdgcn_sched_barrier(0);
                ZeroPairs zp1; // (devmath.hpp mulhi_apx2: the first round's own pairs, written here, dead after it)
                if constexpr (SEALHIP_NTT_INV_FIRST_APX2 && LZ == 1 && kInvApx2)
                    zp1.init();
                FirstPipe<T, 0, LZ, FinalStage<T>::PIPE && !(DY && kLazy<LZ>) && !(WHOLE && LZ == 0)>::run(x, tg0, tw, gbase + jloc, N, neg_p,
                                                                                                         two_p, rdp, zp1);
            }
            const int jb3 = gbase + Arr<T, 3>::tid_index(fresh_tid(wave_base));
            u64 w0[kIL], ws0[kIL];
            RoundStageInv<T, 3, false, 0, LZ>::load(w0, ws0, tw, jb3, N); // lands while the exchange runs
            __builtin_amdgcn_sched_barrier(0);
            h_exchange<T, 4, 3>(x, lds, fresh_tid(wave_base));
            ZeroPairs zp; // (devmath.hpp mulhi_apx2; written again where each phase starts)
            if constexpr (kLazy<LZ> && kInvApx2)
                zp.init();
            RoundPipeInv<T, 3, false, LZ>::run(x, w0, ws0, tw, jb3, N, two_p, neg_p, rdp, zp);
            const int jb2 = gbase + Arr<T, 2>::tid_index(fresh_tid(wave_base));
            RoundStageInv<T, 2, false, 0, LZ>::load(w0, ws0, tw, jb2, N);
            __builtin_amdgcn_sched_barrier(0);
            h_exchange<T, 3, 2>(x, lds, fresh_tid(wave_base));
            if constexpr (kLazy<LZ> && kInvApx2)
                zp.init();
            RoundPipeInv<T, 2, false, LZ>::run(x, w0, ws0, tw, jb2, N, two_p, neg_p, rdp, zp);
            RoundStageInv<T, 1, true, 0, LZ>::load(w0, ws0, tw, gbase, N); // block-uniform twiddles -> scalar loads
            h_exchange<T, 2, 1>(x, lds, fresh_tid(wave_base));
            if constexpr (kLazy<LZ> && kInvApx2)
                zp.init();
            if constexpr (WHOLE)
            {
                RoundPipeInv<T, 1, true, LZ, 0, 3>::run(x, w0, ws0, tw, gbase, N, two_p, neg_p, rdp, zp);
                // the row's top layer: slot bit 4 of arrangement 1 is index bit T - 1
                if constexpr (FP)
                {
                    // (inputs at most 4p in magnitude)
                    const double pd = fp_of(two_p), pinv = fp_of(neg_p);
                    const double c_sum = static_cast<double>(P.inv_n), c_diff = static_cast<double>(P.inv_n_w);
#pragma unroll
                    for (int s2 = 0; s2 < 16; s2++)
                    {
                        const double u = fp_of(x[s2]), v = fp_of(x[s2 | 16]);
                        x[s2] = fp_bits(fp_mulmod(u + v, c_sum, pd, pinv));
                        x[s2 | 16] = fp_bits(fp_mulmod(u - v, c_diff, pd, pinv));
                    }
                }
                else
                {
                    // BackwardLazyLast as ntt_inv_top_kernel applies it; with lazy sums the operands are below
                    // 2^shift(T - 1) p, so that multiple of p keeps the difference non-negative
                    const u64 addend = kLazy<LZ> ? (0 - neg_p) << InvLazy<T, LZ>::shift(T - 1) : two_p;
#pragma unroll
                    for (int s2 = 0; s2 < 16; s2++)
                    {
                        const u64 u = x[s2], v = x[s2 | 16];
                        u64 tt = u + v;
                        if (LZ == 0)
                            tt = tt >= two_p ? tt - two_p : tt;
                        u64 a0 = mulmod_lazy_hs<true>(tt, P.inv_n, P.inv_n_shoup, neg_p);
                        u64 a1 = mulmod_lazy_hs<true>(u - v + addend, P.inv_n_w, P.inv_n_w_shoup, neg_p);
                        if (canonical)
                        {
                            a0 = a0 >= p ? a0 - p : a0;
                            a1 = a1 >= p ? a1 - p : a1;
                        }
                        x[s2] = a0;
                        x[s2 | 16] = a1;
                    }
                }
            }
            else
                RoundPipeInv<T, 1, true, LZ>::run(x, w0, ws0, tw, gbase, N, two_p, neg_p, rdp, zp);
            {
                const int jb = Arr<T, 1>::tid_index(fresh_tid(wave_base));
#pragma unroll
                for (int s = 0; s < 32; s += 2)
                {
                    if constexpr (FP) // canonical residues: below 2p as the consumers of the lazy form expect, and exact
                        store_nt(halfp + jb + Arr<T, 1>::slot_index(s), fp_to_u64(fp_canonical(fp_of(x[s]), fp_of(two_p), fp_of(neg_p))),
                                 fp_to_u64(fp_canonical(fp_of(x[s + 1]), fp_of(two_p), fp_of(neg_p))));
                    else
                        store_nt(halfp + jb + Arr<T, 1>::slot_index(s), x[s], x[s + 1]);
                }
            }
        }

        // top inverse layer (gap N/2): x0 = (u+v)*n^-1, x1 = (u-v+2p)*(w*n^-1)  (BackwardLazyLast, ntt.cpp:274-281)
        __global__ __launch_bounds__(256) void ntt_inv_top_kernel(u64 *__restrict__ data,
                                                                  const PrimeDev *__restrict__ primes, RowMap map,
                                                                  int logn, std::size_t npairs, int flags)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t half = static_cast<std::size_t>(1) << (logn - 1);
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < npairs; i += stride)
            {
                // i enumerates 16-byte pairs of the lower halves: row = i / (N/4), pair inside the half = i % (N/4)
                const std::size_t row = i >> (logn - 2);
                const std::size_t off = (i & ((half >> 1) - 1)) * 2;
                const unsigned short pid = map.prime[row % map.rows];
                if (pid == kSkipRow)
                    continue;
                const PrimeDev &P = primes[pid];
                const u64 p = P.p, two_p = P.two_p;
                u64 *lo = data + (row << logn) + off;
                u64 *hi = lo + half;
                const ulonglong2 a = *reinterpret_cast<const ulonglong2 *>(lo);
                const ulonglong2 b = *reinterpret_cast<const ulonglong2 *>(hi);
                ulonglong2 r0, r1;
                u64 t0 = a.x + b.x, t1 = a.y + b.y;
                t0 = t0 >= two_p ? t0 - two_p : t0;
                t1 = t1 >= two_p ? t1 - two_p : t1;
                r0.x = mulmod_lazy(t0, P.inv_n, P.inv_n_shoup, p);
                r0.y = mulmod_lazy(t1, P.inv_n, P.inv_n_shoup, p);
                r1.x = mulmod_lazy(a.x - b.x + two_p, P.inv_n_w, P.inv_n_w_shoup, p);
                r1.y = mulmod_lazy(a.y - b.y + two_p, P.inv_n_w, P.inv_n_w_shoup, p);
                if (flags & kNttCanonical)
                {
                    r0.x = r0.x >= p ? r0.x - p : r0.x;
                    r0.y = r0.y >= p ? r0.y - p : r0.y;
                    r1.x = r1.x >= p ? r1.x - p : r1.x;
                    r1.y = r1.y >= p ? r1.y - p : r1.y;
                }
                *reinterpret_cast<ulonglong2 *>(lo) = r0;
                *reinterpret_cast<ulonglong2 *>(hi) = r1;
            }
        }

        // The two layers a quarter-row inverse leaves undone, as one streaming pass (N = 2^16): the layer on index bit logn - 2
        // (gap N/4: BackwardLazy, ntt.cpp:265-272, twiddles (N + j) >> (logn - 1) = entries 2 and 3 of the table for the lower and
        // the upper pair) and the top layer (BackwardLazyLast with n^-1 folded in, :274-281), on the four words
        // (j, j + N/4, j + N/2, j + 3N/4) of a row -- the reference's operations in the reference's order, so the `_lazy`
        // entry keeps its representatives. One lane per 16-byte pair of the first quarter.
        __global__ __launch_bounds__(256) void ntt_inv_top2_kernel(u64 *__restrict__ data, const PrimeDev *__restrict__ primes,
                                                                   RowMap map, int logn, std::size_t nitems, int flags)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t quarter = static_cast<std::size_t>(1) << (logn - 2);
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < nitems; i += stride)
            {
                const std::size_t row = i >> (logn - 3); // N/8 pairs per row
                const std::size_t off = (i & ((quarter >> 1) - 1)) * 2;
                const unsigned short pid = map.prime[row % map.rows];
                if (pid == kSkipRow)
                    continue;
                const PrimeDev &P = primes[pid];
                const u64 p = P.p, two_p = P.two_p;
                const u64x2 WA = ((tw_global_t)P.inv)[2], WB = ((tw_global_t)P.inv)[3];
                u64 *q0 = data + (row << logn) + off;
                ulonglong2 x[4], y[4];
#pragma unroll
                for (int r = 0; r < 4; r++)
                    x[r] = *reinterpret_cast<const ulonglong2 *>(q0 + r * quarter);
                const auto lazy_pair = [&](u64 u, u64 v, u64 w, u64 ws, u64 &s, u64 &d) { // BackwardLazy
                    u64 tt = u + v;
                    s = tt >= two_p ? tt - two_p : tt;
                    d = mulmod_lazy(u - v + two_p, w, ws, p);
                };
                const auto last_pair = [&](u64 u, u64 v, u64 &lo, u64 &hi) { // BackwardLazyLast
                    u64 tt = u + v;
                    tt = tt >= two_p ? tt - two_p : tt;
                    lo = mulmod_lazy(tt, P.inv_n, P.inv_n_shoup, p);
                    hi = mulmod_lazy(u - v + two_p, P.inv_n_w, P.inv_n_w_shoup, p);
                    if (flags & kNttCanonical)
                    {
                        lo = lo >= p ? lo - p : lo;
                        hi = hi >= p ? hi - p : hi;
                    }
                };
                u64 s01, d01, s23, d23;
                lazy_pair(x[0].x, x[1].x, WA.x, WA.y, s01, d01);
                lazy_pair(x[2].x, x[3].x, WB.x, WB.y, s23, d23);
                last_pair(s01, s23, y[0].x, y[2].x);
                last_pair(d01, d23, y[1].x, y[3].x);
                lazy_pair(x[0].y, x[1].y, WA.x, WA.y, s01, d01);
                lazy_pair(x[2].y, x[3].y, WB.x, WB.y, s23, d23);
                last_pair(s01, s23, y[0].y, y[2].y);
                last_pair(d01, d23, y[1].y, y[3].y);
#pragma unroll
                for (int r = 0; r < 4; r++)
                    *reinterpret_cast<ulonglong2 *>(q0 + r * quarter) = y[r];
            }
        }

        // rows a launch really transforms (rows mapped to kSkipRow are left alone): the unit count of the profiler
        inline double transformed_rows(std::size_t nrows, const RowMap &map)
        {
            int live = 0;
            for (int r = 0; r < map.rows; r++)
                live += map.prime[r] != kSkipRow;
            return static_cast<double>(nrows / map.rows) * live;
        }

        // A launch whose live rows mix primes below 2^50 with larger ones (the usual CKKS chain: 60-bit ends, 40-bit middle)
        // is issued as two: the rows that can take the floating-point instance, then the others. -> true when both kinds
        // are live; `fp_rows` / `rest` are the map with the other kind masked out.
        inline bool split_by_fp(const Engine &e, const RowMap &map, RowMap &fp_rows, RowMap &rest)
        {
            fp_rows = rest = map;
            int n_fp = 0, n_rest = 0;
            for (int r = 0; r < map.rows; r++)
            {
                if (map.prime[r] == kSkipRow)
                    continue;
                if (e.tables[map.prime[r]].p < kFpPrimeBound)
                {
                    rest.prime[r] = kSkipRow;
                    n_fp++;
                }
                else
                {
                    fp_rows.prime[r] = kSkipRow;
                    n_rest++;
                }
            }
            return n_fp > 0 && n_rest > 0;
        }
        inline bool fp64_enabled()
        {
            static const bool off = std::getenv("SEALHIP_NTT_NO_FP64") != nullptr;
            return !off;
        }

        template <int LOGN>
        hipError_t launch_half_inv(const Engine &e, u64 *data, std::size_t nrows, const RowMap &map, int flags,
                                   const u64 *src = nullptr, std::size_t src_poly_stride = 0,
                                   const DyadicSrc *dyadic = nullptr)
        {
            constexpr int T = LOGN - 1;
            if (fp64_enabled() && (flags & (kNttAnyRep | kNttCanonical)) != 0 && !dyadic)
            {
                RowMap a, b;
                if (split_by_fp(e, map, a, b))
                {
                    const hipError_t err = launch_half_inv<LOGN>(e, data, nrows, a, flags, src, src_poly_stride);
                    return err != hipSuccess ? err : launch_half_inv<LOGN>(e, data, nrows, b, flags, src, src_poly_stride);
                }
            }
            const std::size_t lds_bytes = static_cast<std::size_t>(hpad(1 << (T - 1))) * 8;
            if (nrows % map.rows != 0)
                return hipErrorInvalidValue;
            const LiveSlots live = live_slots(map);
            if (live.n == 0)
                return hipSuccess;
            const std::size_t chunk = ((nrows / map.rows) * live.n + 7) / 8; // live rows per XCD
            const std::size_t blocks = chunk * 16;
            if (blocks > 0x7fffffffull)
                return hipErrorInvalidValue;
            bool did_quarter = false;
            {
                ProfScope prof(e, "ntt_inv_half", transformed_rows(nrows, map));
                // lazy-sum schedule (InvLazy): the stored values keep their residue class and stay below 2p, but not the
                // reference's representative -- only where the caller says so (kNttAnyRep: its inputs are below 2p and
                // the consuming kernel canonicalises) and no live prime can wrap
                // (the canonicalising wrapper, ntt.h:328-333, erases the representative too: its inputs are what the reference
                //  itself requires of an inverse transform, values below 2p)
                static const bool exact_only = std::getenv("SEALHIP_NTT_EXACT_INV") != nullptr;
                bool lazy = (flags & (kNttAnyRep | kNttCanonical)) != 0 && !exact_only;
                bool dense = lazy; // (round 4) the dense schedule where the sparse one has no head-room: primes of 2^45 .. 2^60
                for (int i = 0; dense && i < live.n; i++)
                    dense = bounds::inv_dense_admits(kInvLayers<LOGN>, e.tables[map.prime[live.slot[i]]].p);
                for (int i = 0; lazy && i < live.n; i++)
                    lazy = bounds::inv_lazy_admits(kInvLayers<LOGN>, e.tables[map.prime[live.slot[i]]].p);
                dense = dense && !lazy;
                // floating-point instance: same contract (inputs below 2p, any representative out), every live prime below 2^50
                bool fp = (flags & (kNttAnyRep | kNttCanonical)) != 0 && fp64_enabled() && !dyadic;
                for (int i = 0; fp && i < live.n; i++)
                    fp = e.tables[map.prime[live.slot[i]]].p < kFpPrimeBound;
                const DyadicSrc dy = dyadic ? *dyadic : DyadicSrc{};
                // (A one-launch standalone inverse by sibling hand-off -- the second finisher of a row applying the top layer to
                //  both halves -- was measured in round 2 and brought nothing; its cross-workgroup publish
                //  rested on workgroup-scope fences, so the path was removed rather than kept as an unsupported option.)
                if constexpr (LOGN <= 15)
                {
                    // whole-row form (see the kernel): standalone floating-point transforms (the top layer is not left to a
                    // consumer). Bit log n of SEALHIP_NTT_WHOLE_ROW (default: 2^14 and 2^15).
                    static const unsigned long whole_mask = [] {
                        const char *env = exp_env("SEALHIP_NTT_WHOLE_ROW");
                        return env ? std::strtoul(env, nullptr, 0) : ((1ul << 14) | (1ul << 15));
                    }();
                    if (!dyadic && !(flags & kNttDeferTop) && ((whole_mask >> LOGN) & 1))
                    {
                        const std::size_t wlds = static_cast<std::size_t>(hpad(1 << (LOGN - 1))) * 8;
                        const int canon = (flags & kNttCanonical) ? 1 : 0;
#define SEALHIP_INV_WHOLE(LZ_)                                                                                          \
    ntt_inv_half_kernel<LOGN + 1, LZ_, false, true>                                                                      \
        <<<static_cast<unsigned>(chunk * 8), 1 << (LOGN - 5), wlds, e.lane().stream>>>(                                  \
            data, e.d_primes, map, nrows, chunk, src, src_poly_stride, live, dy, canon)
                        // (the lazy-sum schedule of the larger shape has one more layer: its own bound on the primes -- the
                        //  predicate takes the layer count of the instance that is launched, ntt_inv_half_kernel<LOGN + 1, ..>)
                        bool lazy_w = lazy, dense_w = (lazy || dense);
                        for (int i = 0; lazy_w && i < live.n; i++)
                            lazy_w = bounds::inv_lazy_admits(kInvLayers<LOGN + 1>, e.tables[map.prime[live.slot[i]]].p);
                        for (int i = 0; dense_w && i < live.n; i++)
                            dense_w = bounds::inv_dense_admits(kInvLayers<LOGN + 1>, e.tables[map.prime[live.slot[i]]].p);
                        if (fp)
                            SEALHIP_INV_WHOLE(2);
                        else if (lazy_w)
                            SEALHIP_INV_WHOLE(1);
                        else if (dense_w)
                            SEALHIP_INV_WHOLE(3);
                        else
                            SEALHIP_INV_WHOLE(0);
#undef SEALHIP_INV_WHOLE
                        return hipGetLastError();
                    }
                }
                if constexpr (LOGN == 16)
                {
                    // (round 4) standalone transforms at N = 2^16: quarter-row workgroups of the N = 2^15 shape (two per CU) and one
                    // streaming radix-4 pass for the two top layers, instead of the 1024-lane half-row kernel (one per CU) and
                    // the streaming top-layer pass. SEALHIP_NTT_QUARTER=0 (measurement build) restores the latter for A/B.
                    static const bool quarter_off = exp_env("SEALHIP_NTT_QUARTER") != nullptr && std::atoi(exp_env("SEALHIP_NTT_QUARTER")) == 0;
                    if (!dyadic && !(flags & kNttDeferTop) && !quarter_off)
                    {
                        constexpr int QL = LOGN - 1; // the shape: ntt_inv_half_kernel<15, ..>, 14 on-chip layers
                        const std::size_t qlds = static_cast<std::size_t>(hpad(1 << (QL - 2))) * 8;
                        const std::size_t qblocks = chunk * 32; // four workgroups per live row, eight XCDs
                        if (qblocks > 0x7fffffffull)
                            return hipErrorInvalidValue;
                        bool lazy_q = (flags & (kNttAnyRep | kNttCanonical)) != 0 && !exact_only, dense_q = lazy_q;
                        for (int i = 0; lazy_q && i < live.n; i++)
                            lazy_q = bounds::inv_lazy_admits(kInvLayers<QL>, e.tables[map.prime[live.slot[i]]].p);
                        for (int i = 0; dense_q && i < live.n; i++)
                            dense_q = bounds::inv_dense_admits(kInvLayers<QL>, e.tables[map.prime[live.slot[i]]].p);
#define SEALHIP_INV_QUARTER(LZ_)                                                                                          \
    ntt_inv_half_kernel<QL, LZ_, false, false, true>                                                                        \
        <<<static_cast<unsigned>(qblocks), 1 << (QL - 6), qlds, e.lane().stream>>>(data, e.d_primes, map, nrows, chunk, src, \
                                                                                    src_poly_stride, live, dy)
                        if (fp)
                            SEALHIP_INV_QUARTER(2);
                        else if (lazy_q)
                            SEALHIP_INV_QUARTER(1);
                        else if (dense_q)
                            SEALHIP_INV_QUARTER(3);
                        else
                            SEALHIP_INV_QUARTER(0);
#undef SEALHIP_INV_QUARTER
                        hipError_t qerr = hipGetLastError();
                        if (qerr != hipSuccess)
                            return qerr;
                        did_quarter = true;
                    }
                }
                if (!did_quarter)
                {
#define SEALHIP_INV_HALF(LZ_, DY_)                                                                                    \
    ntt_inv_half_kernel<LOGN, LZ_, DY_><<<static_cast<unsigned>(blocks), 1 << (LOGN - 6), lds_bytes, e.lane().stream>>>( \
        data, e.d_primes, map, nrows, chunk, src, src_poly_stride, live, dy)
                if (dyadic)
                {
                    if (lazy)
                        SEALHIP_INV_HALF(1, true);
                    else if (dense)
                        SEALHIP_INV_HALF(3, true);
                    else
                        SEALHIP_INV_HALF(0, true);
                }
                else if (fp)
                    SEALHIP_INV_HALF(2, false);
                else if (lazy)
                    SEALHIP_INV_HALF(1, false);
                else if (dense)
                    SEALHIP_INV_HALF(3, false);
                else
                    SEALHIP_INV_HALF(0, false);
#undef SEALHIP_INV_HALF
                hipError_t err = hipGetLastError();
                if (err != hipSuccess)
                    return err;
                }
            }
            if (did_quarter)
            {
                const std::size_t nitems = nrows << (LOGN - 3);
                std::size_t grid = (nitems + 255) / 256;
                if (grid > 256u * 32u)
                    grid = 256u * 32u;
                ProfScope prof(e, "ntt_inv_top", 0);
                ntt_inv_top2_kernel<<<static_cast<unsigned>(grid), 256, 0, e.lane().stream>>>(data, e.d_primes, map, LOGN, nitems, flags);
                return hipGetLastError();
            }
            if (flags & kNttDeferTop)
                return hipSuccess; // the consumer applies the top layer (and the canonicalisation) on load
            const std::size_t npairs = nrows << (LOGN - 2);
            std::size_t grid = (npairs + 255) / 256;
            if (grid > 256u * 32u)
                grid = 256u * 32u;
            ProfScope prof(e, "ntt_inv_top", transformed_rows(nrows, map));
            ntt_inv_top_kernel<<<static_cast<unsigned>(grid), 256, 0, e.lane().stream>>>(data, e.d_primes, map, LOGN, npairs,
                                                                               flags);
            return hipGetLastError();
        }

        template <int LOGN>
        hipError_t launch_half(const Engine &e, u64 *data, std::size_t nrows, const RowMap &map, int flags,
                               const NttSource &src)
        {
            constexpr int T = LOGN - 1;
            std::size_t lds_bytes = static_cast<std::size_t>(hpad(1 << (T - 1))) * 8;
#ifdef SEALHIP_NTT_EXPERIMENT
            if (const char *ex = std::getenv("SEALHIP_NTT_LDS_EXTRA")) // lower the occupancy on purpose
            {
                lds_bytes += std::strtoul(ex, nullptr, 0);
                (void)hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 0>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds_bytes));
            }
#endif
            if (nrows % map.rows != 0)
                return hipErrorInvalidValue;
            if (fp64_enabled() && (flags & (kNttAnyRep | kNttCanonical)) != 0 && (flags & kNttReduceOut) == 0 && !src.base[0])
            {
                RowMap a, b; // (in-place launches only: a gathered launch's sources are not known per row here)
                if (split_by_fp(e, map, a, b))
                {
                    const hipError_t err = launch_half<LOGN>(e, data, nrows, a, flags, src);
                    return err != hipSuccess ? err : launch_half<LOGN>(e, data, nrows, b, flags, src);
                }
            }
            const LiveSlots live = live_slots(map);
            if (live.n == 0)
                return hipSuccess;
            const std::size_t chunk = ((nrows / map.rows) * live.n + 7) / 8; // live rows per XCD
            const std::size_t blocks = chunk * 16;
            if (blocks > 0x7fffffffull)
                return hipErrorInvalidValue;
            // STRICT mode means "no wrap-around": Harvey's corrected butterflies (one conditional subtraction each)
            // guarantee it for any prime. Where the consumer takes any representative (kNttAnyRep, kNttApprox) or the
            // canonical residue is what is returned (kNttCanonical), and every live prime leaves the head-room that the cheaper
            // schedules are proved on (ntt_bounds.hpp section 2: nothing can wrap there either), the residues are the same and
            // the flag is dropped for the launch (round 4). The `_lazy` entries and the 60-bit rows keep the corrected sequence.
            if ((flags & kNttStrict) != 0 && (flags & (kNttAnyRep | kNttCanonical | kNttApprox)) != 0 && (flags & kNttReduceOut) == 0)
            {
                static const bool exact_fwd = std::getenv("SEALHIP_NTT_EXACT_FWD") != nullptr;
                bool ok = !exact_fwd;
                for (int i = 0; ok && i < live.n; i++)
                    ok = bounds::fwd_canon_admits(e.tables[map.prime[live.slot[i]]].p, LOGN);
                if (ok)
                    flags &= ~kNttStrict;
            }
            // STRICT launches on primes without the head-room of the rule above (the 60-bit Bsk rows of a BFV multiply) whose
            // consumer takes any representative below 2p (kNttReduceOut | kNttAnyRep): the dense lazy schedule of
            // ntt_bounds.hpp section 2b instead of a conditional subtraction per butterfly -- the reference's own butterfly,
            // every word brought back below 2p before rounds 2 and 3 and in the store. Same residues, nothing wraps.
            bool dense = SEALHIP_NTT_FWD_DENSE_DEFAULT && (flags & kNttStrict) != 0 && (flags & kNttReduceOut) != 0 && (flags & kNttAnyRep) != 0 &&
                         (flags & kNttCanonical) == 0 && !src.base[0] && std::getenv("SEALHIP_NTT_EXACT_FWD") == nullptr;
            for (int i = 0; dense && i < live.n; i++)
                dense = bounds::fwd_dense_admits(e.tables[map.prime[live.slot[i]]].p, LOGN);
            // SEALHIP_NTT_NO_TICKET=1 (A/B of the hand-off cost) re-opens the race: read by the measurement-only build alone
            // (engine.hpp exp_env); in the shipping library this is the constant false
            static const bool no_ticket = exp_env("SEALHIP_NTT_NO_TICKET") != nullptr;
            // (kNttTopDone: no workgroup reads the other's half, nothing to hand off)
            const bool top_done = (flags & kNttTopDone) != 0;
            // (a STRICT launch may start below the top layer only on the dense schedule: Harvey's sequence has no such instance)
            if (top_done && (((flags & (kNttReduceOut | kNttStrict | kNttCanonical)) != kNttReduceOut && !dense) || src.base[0]))
                return hipErrorInvalidValue;
            // a launch whose live rows are all gathered from another buffer writes no row that anybody reads: nothing to
            // hand off either (SEALHIP_NTT_GATHER_TICKET=1 keeps the hand-off, for A/B)
            static const bool gather_ticket = exp_env("SEALHIP_NTT_GATHER_TICKET") != nullptr;
            bool all_gathered = src.base[0] != nullptr && !gather_ticket;
            for (int i = 0; all_gathered && i < live.n; i++)
                all_gathered = src.code[live.slot[i]] != kSkipRow;
            // SEALHIP_NTT_XCHG_TOP builds: the workgroups of a row share the top layer (h_load_top_xchg) wherever the load is a
            // plain or single-prime mod-up one; a flag word per (row, wave) then
            const int red0 = src.base[0] ? src.reduce_mode : 0;
            const bool xchg = SEALHIP_NTT_XCHG_TOP && !top_done && red0 <= 2; // (every launch of the instances with REDUCE <= 3)
            if (xchg)
                flags |= kNttXchgTop;
            const bool no_handoff = !xchg && (no_ticket || top_done || all_gathered);
            unsigned *tickets = no_handoff ? nullptr : e.ntt_tickets(xchg ? nrows * 16 : nrows); // zeroed for this launch, stream-ordered
            if (e.ntt_suppress_signal)
                flags |= kNttDebugNoSignal; // sealhip_debug_ntt_handoff: drive the time-out path
            {
                // (see half_block_map): bit 0 = FP64 digit launches, bit 1 = integer ones
                static const int poly_major = [] {
                    const char *env = exp_env("SEALHIP_NTT_POLY_MAJOR");
                    return env ? std::atoi(env) : 2;
                }();
                bool one_source = poly_major && all_gathered && src.reduce_mode <= 2 && live.n > 1;
                for (int i = 1; one_source && i < live.n; i++) // every live row gathers the same source row (a key-switch digit)
                    one_source = ((src.code[live.slot[i]] ^ src.code[live.slot[0]]) & ~kSrcReduce) == 0;
                if (one_source)
                    flags |= kNttPolyMajorRequest;
            }
            if (!tickets && !no_handoff)
                return hipErrorOutOfMemory;
#ifdef SEALHIP_NTT_EXPERIMENT
            if (const char *sk = std::getenv("SEALHIP_NTT_SKIP"))
                flags |= static_cast<int>(std::strtol(sk, nullptr, 0));
            unsigned long long *trace = nullptr;
            const char *trace_path = std::getenv("SEALHIP_NTT_TRACE");
            if (trace_path)
            {
                flags |= 0x2000;
                if (hipMalloc(&trace, blocks * 64) != hipSuccess || hipMemset(trace, 0, blocks * 64) != hipSuccess ||
                    hipMemcpyToSymbol(HIP_SYMBOL(g_ntt_trace), &trace, sizeof(trace)) != hipSuccess)
                    return hipErrorOutOfMemory;
            }
            struct TraceDump
            {
                unsigned long long *trace;
                const char *path;
                std::size_t blocks;
                hipStream_t stream;
                ~TraceDump()
                {
                    if (!trace)
                        return;
                    (void)hipStreamSynchronize(stream);
                    std::vector<unsigned long long> h(blocks * 8);
                    (void)hipMemcpy(h.data(), trace, blocks * 64, hipMemcpyDeviceToHost);
                    if (FILE *f = std::fopen(path, "wb"))
                    {
                        std::fwrite(h.data(), 8, h.size(), f);
                        std::fclose(f);
                    }
                    (void)hipFree(trace);
                }
            } trace_dump{trace, trace_path, blocks, e.lane().stream};
#endif
            // Floating-point instance (devmath.hpp): every live prime below 2^50, inputs below 2^52 (residues, or gathered
            // words of another key prime, or the output of a load treatment), and a launch that does not ask for the
            // integer sequence's own representatives (canonical output, or a consumer that reduces whatever it reads).
            // The result is the canonical residue, so the integer instances' flags play no further role.
            bool fp = fp64_enabled() && (flags & (kNttAnyRep | kNttCanonical)) != 0 && (flags & kNttReduceOut) == 0;
            for (int i = 0; fp && i < live.n; i++)
                fp = e.tables[map.prime[live.slot[i]]].p < kFpPrimeBound;
            if (fp && src.base[0])
            {
                if (src.reduce_mode == 4 || src.reduce_mode == 5 || src.reduce_mode == 7)
                    fp = src.aux_p < (src.reduce_mode == 4 ? bounds::kFpInputBound : kFpPrimeBound); // (5, 7: sums of two words below 2P)
                else
                    for (std::size_t i = 0; fp && i < e.key_moduli.size(); i++)
                        fp = e.key_moduli[i] < bounds::kFpInputBound;
            }
            if (flags & kNttPolyMajorRequest)
            {
                static const int poly_major = exp_env("SEALHIP_NTT_POLY_MAJOR") ? std::atoi(exp_env("SEALHIP_NTT_POLY_MAJOR")) : 2;
                static const int group = exp_env("SEALHIP_NTT_POLY_GROUP") ? std::atoi(exp_env("SEALHIP_NTT_POLY_GROUP")) : 4;
                flags &= ~kNttPolyMajorRequest;
                if (poly_major & (fp ? 1 : 2))
                {
                    const int g = group > 0 && group < live.n ? group : live.n;
                    flags |= kNttPolyMajor | (g << 16);
                }
            }
            if (flags & kNttAnyRep)
            {
                // the last layer may keep its first operand unreduced only if the grown values cannot wrap
                // (below (2 log n + 3) p < 2^64 for p < 2^58) and nothing expects the [0, 4p) output range
                static const bool exact_only = std::getenv("SEALHIP_NTT_EXACT_FWD") != nullptr;
                bool ok = !exact_only && (flags & (kNttCanonical | kNttStrict)) == 0;
                for (int i = 0; ok && i < live.n; i++)
                    ok = bounds::fwd_lazy_admits(e.tables[map.prime[live.slot[i]]].p, LOGN);
                if (!ok)
                    flags &= ~kNttAnyRep;
            }
            ProfScope prof(e, "ntt_fwd_half", transformed_rows(nrows, map));
#define SEALHIP_FWD_HALF(STRICT_, RED_)                                                                              \
    ntt_fwd_half_kernel<LOGN, STRICT_, RED_><<<static_cast<unsigned>(blocks), 1 << (LOGN - 6), lds_bytes, e.lane().stream>>>( \
        data, e.d_primes, map, nrows, flags, tickets, e.lane().d_fault, e.ntt_spin_limit, src, chunk, live)
            int red = src.base[0] ? src.reduce_mode : 0;
            if (flags & kNttReduceOut)
            {
                if (red != 0 || (flags & kNttCanonical))
                    return hipErrorInvalidValue; // an in-place, non-canonical launch option
                red = top_done ? 6 : 3;
            }
            // butterfly mode 2 (approximate Shoup quotient, one multiplier instruction less per butterfly): the product then
            // lies in [0, 3p), every layer adds 3p instead of 2p and the outputs are below 50p (kNttAnyRep) or 5p. Only where
            // the consumer reduces whatever representative it reads, nothing expects the [0, 4p) range (no canonicalising
            // wrapper, no kNttReduceOut) and 50p cannot wrap: every live prime below 2^58.
            static const bool no_apx = std::getenv("SEALHIP_NTT_EXACT_FWD") != nullptr;
            bool apx = !no_apx && (flags & kNttApprox) != 0 && (flags & (kNttStrict | kNttCanonical | kNttReduceOut)) == 0 &&
                       red != 4 && red != 5 && red != 7;
            for (int i = 0; apx && i < live.n; i++)
                apx = bounds::fwd_lazy_admits(e.tables[map.prime[live.slot[i]]].p, LOGN);
            // The canonicalising wrapper (ntt.h:225-246) on primes with head-room: the canonical residue does not depend on
            // the representatives the layers pass on, so an in-place canonical transform runs the cheapest exact schedule --
            // approximate quotient, no Barrett step in the last layer, values below (4 + g log n) p for inputs below 4p
            // (ntt_bounds.hpp section 2: fwd_canon_admits) -- and canonicalises with one reduction as it stores.
            // SEALHIP_NTT_CANON_EXACT=1: the reference's sequence.
            static const bool canon_exact = std::getenv("SEALHIP_NTT_CANON_EXACT") != nullptr;
            bool capx = !no_apx && !canon_exact && !fp && red == 0 && (flags & kNttCanonical) != 0 &&
                        (flags & (kNttStrict | kNttReduceOut)) == 0;
            for (int i = 0; capx && i < live.n; i++) // (inputs below 4p, the range include/sealhip.h documents)
                capx = bounds::fwd_canon_admits(e.tables[map.prime[live.slot[i]]].p, LOGN);
            if (capx)
            {
                apx = true;
                flags |= kNttAnyRep;
                bool sq = exp_env("SEALHIP_NTT_CANON_BARRETT") == nullptr; // (A/B: the Barrett step of round 3)
                for (int i = 0; sq && i < live.n; i++)
                    sq = bounds::small_quot_admits(e.tables[map.prime[live.slot[i]]].p, bounds::fwd_canon_output_mult(LOGN));
                if (sq)
                    flags |= kNttSmallQuot;
            }
            if (red == 7 && (!fp || !kStoreExchange<T, 3, 7>))
                return hipErrorInvalidValue; // ntt_can_fuse_moddown said no: the caller runs moddown_post itself
            if (fp && (red == 4 || red == 5 || red == 7))
            {
                NttSource fsrc = src; // the special prime's constants as doubles (h_load_top)
                const double P = static_cast<double>(src.aux_p), Pinv = 1.0 / P;
                const double c0 = static_cast<double>(src.aux_top[0]), c2 = static_cast<double>(src.aux_top[2]);
                std::memcpy(&fsrc.aux_p, &P, 8);
                std::memcpy(&fsrc.aux_cr1, &Pinv, 8);
                std::memcpy(&fsrc.aux_top[0], &c0, 8);
                std::memcpy(&fsrc.aux_top[2], &c2, 8);
                if (red == 7)
                {
                    if constexpr (kStoreExchange<T, 3, 7>)
                        ntt_fwd_half_kernel<LOGN, 3, 7><<<static_cast<unsigned>(blocks), 1 << (LOGN - 6), lds_bytes, e.lane().stream>>>(
                            data, e.d_primes, map, nrows, flags, tickets, e.lane().d_fault, e.ntt_spin_limit, fsrc, chunk, live);
                }
                else if (red == 5)
                    ntt_fwd_half_kernel<LOGN, 3, 5><<<static_cast<unsigned>(blocks), 1 << (LOGN - 6), lds_bytes, e.lane().stream>>>(
                        data, e.d_primes, map, nrows, flags, tickets, e.lane().d_fault, e.ntt_spin_limit, fsrc, chunk, live);
                else
                    ntt_fwd_half_kernel<LOGN, 3, 4><<<static_cast<unsigned>(blocks), 1 << (LOGN - 6), lds_bytes, e.lane().stream>>>(
                        data, e.d_primes, map, nrows, flags, tickets, e.lane().d_fault, e.ntt_spin_limit, fsrc, chunk, live);
            }
            else if (fp)
            {
                if (red == 2)
                    SEALHIP_FWD_HALF(3, 2);
                else if (red == 1)
                    SEALHIP_FWD_HALF(3, 1);
                else
                    SEALHIP_FWD_HALF(3, 0);
            }
            else if (apx)
            {
                if (red == 2)
                    SEALHIP_FWD_HALF(2, 2);
                else if (red == 1)
                    SEALHIP_FWD_HALF(2, 1);
                else
                    SEALHIP_FWD_HALF(2, 0);
            }
            else if (dense && top_done)
                SEALHIP_FWD_HALF(4, 6);
            else if (dense)
                SEALHIP_FWD_HALF(4, 3);
            else if (flags & kNttStrict)
            {
                if (red == 5)
                    SEALHIP_FWD_HALF(1, 5);
                else if (red == 4)
                    SEALHIP_FWD_HALF(1, 4);
                else if (red == 3)
                    SEALHIP_FWD_HALF(1, 3);
                else if (red == 2)
                    SEALHIP_FWD_HALF(1, 2);
                else if (red == 1)
                    SEALHIP_FWD_HALF(1, 1);
                else
                    SEALHIP_FWD_HALF(1, 0);
            }
            else
            {
                if (red == 6)
                    SEALHIP_FWD_HALF(0, 6);
                else if (red == 5)
                    SEALHIP_FWD_HALF(0, 5);
                else if (red == 4)
                    SEALHIP_FWD_HALF(0, 4);
                else if (red == 3)
                    SEALHIP_FWD_HALF(0, 3);
                else if (red == 2)
                    SEALHIP_FWD_HALF(0, 2);
                else if (red == 1)
                    SEALHIP_FWD_HALF(0, 1);
                else
                    SEALHIP_FWD_HALF(0, 0);
            }
#undef SEALHIP_FWD_HALF
            return hipGetLastError();
        }

        template <int LOGN>
        hipError_t init_half()
        {
            const int lds_bytes = hpad(1 << (LOGN - 2)) * 8;
            hipError_t err = hipSuccess;
            const void *fwd[23] = { reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 4, 3>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 4, 6>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 6>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 5>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 5>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 5>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 0>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 1>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 2>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 4>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 4>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 4>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 3>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 3>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 0>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 1>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 0, 2>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 0>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 1>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 1, 2>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 2, 0>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 2, 1>),
                                    reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 2, 2>) };
            for (const void *f : fwd)
            {
                err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (err != hipSuccess)
                    return err;
            }
            if constexpr (kStoreExchange<LOGN - 1, 3, 7>)
            {
                err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_fwd_half_kernel<LOGN, 3, 7>),
                                          hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (err != hipSuccess)
                    return err;
            }
            const void *inv[7] = { reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 2, false>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 1, false>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 0, false>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 3, false>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 1, true>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 3, true>),
                                   reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 0, true>) };
            for (const void *f : inv)
            {
                err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                if (err != hipSuccess)
                    return err;
            }
            if constexpr (LOGN == 15)
            {
                // the quarter-row instances that serve rings of 2^16 (same shape, same LDS)
                const void *quarter[4] = { reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 2, false, false, true>),
                                           reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 1, false, false, true>),
                                           reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 3, false, false, true>),
                                           reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN, 0, false, false, true>) };
                for (const void *f : quarter)
                {
                    err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, lds_bytes);
                    if (err != hipSuccess)
                        return err;
                }
            }
            if constexpr (LOGN <= 15)
            {
                const void *whole[4] = { reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN + 1, 2, false, true>),
                                         reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN + 1, 1, false, true>),
                                         reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN + 1, 3, false, true>),
                                         reinterpret_cast<const void *>(&ntt_inv_half_kernel<LOGN + 1, 0, false, true>) };
                for (const void *f : whole)
                {
                    err = hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, hpad(1 << (LOGN - 1)) * 8);
                    if (err != hipSuccess)
                        return err;
                }
            }
            return err;
        }

        void make_rounds(NttPass &ps, int lo, int hi, bool inverse)
        {
            const int count = hi - lo + 1;
            const int first = ((count - 1) % 4) + 1;
            ps.nrounds = 0;
            int done = 0;
            while (done < count)
            {
                const int size = done == 0 ? first : 4;
                int r_lo, r_hi;
                if (!inverse)
                {
                    r_hi = hi - done;
                    r_lo = r_hi - size + 1;
                }
                else
                {
                    r_lo = lo + done;
                    r_hi = r_lo + size - 1;
                }
                NttRound &rd = ps.rounds[ps.nrounds++];
                rd.beta = r_lo < ps.t - 4 ? r_lo : ps.t - 4;
                rd.wlo = r_lo - rd.beta;
                rd.whi = r_hi - rd.beta;
                done += size;
            }
        }
    } // namespace

    NttPlan plan_ntt(int logn, bool inverse, int flags)
    {
        NttPlan plan{};
        plan.logn = logn;
        plan.serial = logn < 4;
        plan.flags = flags;
        if (plan.serial)
        {
            plan.npass = 0;
            return plan;
        }
        auto init = [&](NttPass &ps, int t, int c, int b_lo, int lo, int hi) {
            ps.logn = logn;
            ps.t = t;
            ps.c = c;
            ps.b_lo = b_lo;
            ps.flags = flags & kNttStrict;
            make_rounds(ps, lo, hi, inverse);
        };
        // (round 4: the two-pass split for logn > 13 -- strided columns, then contiguous rows -- went with its switch
        //  SEALHIP_NTT_TWO_PASS: rings of 2^14 .. 2^16 are served by the single-pass kernels below, nothing else reached it)
        if (logn > kTileBitsMax)
            throw std::logic_error("plan_ntt: rings above 2^13 take the single-pass kernels");
        plan.npass = 1;
        init(plan.pass[0], logn, 0, 0, 0, logn - 1);
        plan.pass[plan.npass - 1].flags |= flags & kNttCanonical; // wrapper fused into the last store
        return plan;
    }

    template <int DIR>
    static hipError_t launch_dir(const Engine &e, u64 *data, size_t nrows, const RowMap &map, const NttPlan &plan)
    {
        if (nrows == 0)
            return hipSuccess;
        if (plan.serial)
        {
            const int threads = 64;
            const unsigned blocks = static_cast<unsigned>((nrows + threads - 1) / threads);
            ntt_serial_kernel<DIR><<<blocks, threads, 0, e.lane().stream>>>(data, e.d_primes, map, plan.logn, plan.flags, nrows);
            return hipGetLastError();
        }
        for (int i = 0; i < plan.npass; i++)
        {
            const NttPass &ps = plan.pass[i];
            const int threads = 1 << (ps.t - 4);
            const size_t lds_bytes = (static_cast<size_t>(1) << ps.t) * 8 + (static_cast<size_t>(1) << (ps.t - 4)) * 8;
            const size_t blocks = nrows << (plan.logn - ps.t);
            if (blocks > 0x7fffffffull)
                return hipErrorInvalidValue;
            hipError_t err;
            {
                ProfScope prof(e, DIR == 0 ? "ntt_fwd_pass" : "ntt_inv_pass", transformed_rows(nrows, map));
                ntt_pass_kernel<DIR><<<static_cast<unsigned>(blocks), threads, lds_bytes, e.lane().stream>>>(data, e.d_primes,
                                                                                                      map, ps);
                err = hipGetLastError();
            }
            if (err != hipSuccess)
                return err;
        }
        return hipSuccess;
    }

    // ---- the arithmetic ceiling of a butterfly sequence, measured on the device it runs on (sealhip_debug_butterfly_rate;
    // bench.py's roofline.valu_ceiling). Nothing but the butterflies of the single-pass kernels' rounds: the same lock-step
    // sequences, 32 values and four per-lane twiddles in registers, no loads, no exchanges, the same launch bounds (two
    // workgroups of 512 lanes per CU). KIND 0: the reference's lazy butterfly (exact Shoup quotient), 1 / 2: the approximate
    // quotients of levels 1 / 2, 3: the FP64 butterfly, 4: the lazy-sum inverse butterfly with the level-2 quotient,
    // 5: the inverse butterfly with the reference's sequence.
    namespace
    {
        template <int KIND>
        __global__ __launch_bounds__(512, 4) void butterfly_rate_kernel(u64 *__restrict__ sink, PrimeDev P, int iters)
        {
            u64 x[16], w[kIL], ws[kIL]; // (16 values: the sequence is what is measured, and nothing may spill)
            const u64 seed = (static_cast<u64>(blockIdx.x) * 512 + threadIdx.x) * 0x9E3779B97F4A7C15ull;
            constexpr bool FP = KIND == 3;
#pragma unroll
            for (int i = 0; i < 16; i++)
            {
                const u64 v = (seed + static_cast<u64>(i) * 0xBF58476D1CE4E5B9ull) % P.p;
                x[i] = FP ? fp_bits(fp_from_u64(v)) : v;
            }
#pragma unroll
            for (int j = 0; j < kIL; j++)
            {
                const u64 wv = (seed ^ (0x94D049BB133111EBull * (j + 1))) % P.p;
                w[j] = FP ? fp_bits(fp_from_u64(wv)) : wv;
                ws[j] = static_cast<u64>((static_cast<unsigned __int128>(wv) << 64) / P.p);
            }
            const u64 neg_p = FP ? fp_bits(P.pinv_d) : 0 - P.p, two_p = FP ? fp_bits(P.p_d) : P.two_p;
            ZeroPairs zp;
            zp.init();
            u64 four_p = two_p << 1; // (opaque like fwd_addend's: one v_lshl_add_u64 per second output)
            asm("" : "+s"(four_p));
            for (int it = 0; it < iters; it++)
            {
#pragma unroll
                for (int W = 3; W >= 0; W--)
                {
#pragma unroll
                    for (int c = 0; c < 8; c += kIL)
                    {
                        u64 u[kIL], y[kIL];
                        const int bit = 1 << W;
#pragma unroll
                        for (int j = 0; j < kIL; j++)
                        {
                            const int sl = (((c + j) >> W) << (W + 1)) | ((c + j) & (bit - 1));
                            u[j] = x[sl];
                            y[j] = x[sl | bit];
                        }
                        if constexpr (KIND == 3)
                        {
#pragma unroll
                            for (int j = 0; j < kIL; j++)
                                fp_butterfly_fwd(u[j], y[j], w[j], fp_of(two_p), fp_of(neg_p));
                        }
                        else if constexpr (KIND == 0)
                            butterflies_fwd_hs<false, kIL, 0>(u, y, w, ws, neg_p, two_p);
                        else if constexpr (KIND == 1)
                            butterflies_fwd_hs<false, kIL, 1>(u, y, w, ws, neg_p, two_p - neg_p);
                        else if constexpr (KIND == 2)
                            butterflies_fwd_apx2<false, kIL>(u, y, w, ws, neg_p, four_p, zp.z);
                        else if constexpr (KIND == 4)
                            butterflies_inv_apx2<false, kIL>(u, y, w, ws, neg_p, four_p, zp.z);
                        else
                            butterflies_inv_hs<false, kIL, 0>(u, y, w, ws, neg_p, two_p);
#pragma unroll
                        for (int j = 0; j < kIL; j++)
                        {
                            const int sl = (((c + j) >> W) << (W + 1)) | ((c + j) & (bit - 1));
                            x[sl] = u[j];
                            x[sl | bit] = y[j];
                        }
                    }
                }
                if constexpr (FP)
                    if ((it & 1) == 1) // (one reduction of every value per eight layers: a little more than the real schedule's)
                    {
#pragma unroll
                        for (int i = 0; i < 16; i++)
                            x[i] = fp_bits(fp_reduce(fp_of(x[i]), fp_of(two_p), fp_of(neg_p)));
                    }
            }
            u64 acc = 0;
#pragma unroll
            for (int i = 0; i < 16; i++)
                acc ^= x[i];
            if (acc == 0x0123456789ABCDEFull) // (never: keeps the arithmetic alive)
                sink[0] = acc;
        }
    } // namespace

    hipError_t ntt_butterfly_rate(const Engine &e, int kind, int prime_id, double *butterflies_per_s)
    {
        if (kind < 0 || kind > 5 || prime_id < 0 || prime_id >= static_cast<int>(e.tables.size()))
            return hipErrorInvalidValue;
        PrimeDev P{};
        hipError_t err = hipMemcpy(&P, e.d_primes + prime_id, sizeof(P), hipMemcpyDeviceToHost);
        if (err != hipSuccess)
            return err;
        if (kind == 3 && P.fwd_d == nullptr)
            return hipErrorInvalidValue; // no FP64 instance for this prime
        u64 *sink = nullptr;
        if ((err = hipMalloc(&sink, 8)) != hipSuccess)
            return err;
        hipStream_t st = e.lane().stream;
        hipEvent_t a, b;
        (void)hipEventCreate(&a);
        (void)hipEventCreate(&b);
        const unsigned blocks = 256u * 2u * 4u; // four rounds of resident workgroups per CU slot
        const int iters = 800;
        // the transforms' occupancy: their 68 KB of LDS admit two workgroups (four waves per SIMD) per CU; this kernel uses
        // none and fewer registers, so it asks for the same amount to run under the same cap
        const std::size_t lds = static_cast<std::size_t>(hpad(1 << 13)) * 8;
        const void *fns[6] = { reinterpret_cast<const void *>(&butterfly_rate_kernel<0>), reinterpret_cast<const void *>(&butterfly_rate_kernel<1>),
                               reinterpret_cast<const void *>(&butterfly_rate_kernel<2>), reinterpret_cast<const void *>(&butterfly_rate_kernel<3>),
                               reinterpret_cast<const void *>(&butterfly_rate_kernel<4>), reinterpret_cast<const void *>(&butterfly_rate_kernel<5>) };
        if ((err = hipFuncSetAttribute(fns[kind], hipFuncAttributeMaxDynamicSharedMemorySize, static_cast<int>(lds))) != hipSuccess)
        {
            (void)hipEventDestroy(a);
            (void)hipEventDestroy(b);
            (void)hipFree(sink);
            return err;
        }
        const auto launch = [&](int n) {
            switch (kind)
            {
            case 0: butterfly_rate_kernel<0><<<blocks, 512, lds, st>>>(sink, P, n); break;
            case 1: butterfly_rate_kernel<1><<<blocks, 512, lds, st>>>(sink, P, n); break;
            case 2: butterfly_rate_kernel<2><<<blocks, 512, lds, st>>>(sink, P, n); break;
            case 3: butterfly_rate_kernel<3><<<blocks, 512, lds, st>>>(sink, P, n); break;
            case 4: butterfly_rate_kernel<4><<<blocks, 512, lds, st>>>(sink, P, n); break;
            default: butterfly_rate_kernel<5><<<blocks, 512, lds, st>>>(sink, P, n); break;
            }
        };
        launch(iters / 8); // warm-up (clocks, code)
        (void)hipEventRecord(a, st);
        launch(iters);
        launch(iters);
        (void)hipEventRecord(b, st);
        err = hipStreamSynchronize(st);
        float ms = 0;
        if (err == hipSuccess)
            err = hipEventElapsedTime(&ms, a, b);
        (void)hipEventDestroy(a);
        (void)hipEventDestroy(b);
        (void)hipFree(sink);
        if (err != hipSuccess)
            return err;
        *butterflies_per_s = 2.0 * blocks * 512.0 * iters * 32.0 / (ms / 1e3);
        return hipGetLastError();
    }

    hipError_t ntt_init_kernels()
    {
        const int max_lds = ((1 << kTileBitsMax) + (1 << (kTileBitsMax - 4))) * 8;
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<0>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (err != hipSuccess)
            return err;
        err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<1>),
                                  hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (err == hipSuccess)
            err = init_half<14>();
        if (err == hipSuccess)
            err = init_half<15>();
        if (err == hipSuccess)
            err = init_half<16>();
        return err;
    }

    bool ntt_can_defer_top(const Engine &e, int k)
    {
        return e.use_half_kernel && e.logn >= 14 && e.logn <= 16 && k <= 32;
    }

    bool ntt_can_fuse_moddown(const Engine &e, int k, u64 p_special)
    {
        static const bool off = exp_env("SEALHIP_KS_MODDOWN_STORE_UNFUSED") != nullptr;
        if (off || !fp64_enabled() || !ntt_can_gather(e) || e.logn < 15 || (SEALHIP_NTT_STORE_EXCHANGE & 1) == 0 ||
            p_special >= kFpPrimeBound)
            return false;
        for (int r = 0; r < k; r++)
            if (e.key_moduli[r] >= kFpPrimeBound)
                return false;
        return true;
    }

    // STRICT mode: may a producer apply the forward transform's top layer to rows on these primes (kNttTopDone)? Only the
    // dense lazy schedule has an instance that starts below it (launch_half)
    bool ntt_strict_top_done_ok(const Engine &e, const RowMap &map)
    {
        if (!SEALHIP_NTT_FWD_DENSE_DEFAULT || !e.use_half_kernel || e.logn < 14 || e.logn > 16 || std::getenv("SEALHIP_NTT_EXACT_FWD") != nullptr)
            return false;
        for (int r = 0; r < map.rows; r++)
            if (map.prime[r] != kSkipRow && !bounds::fwd_dense_admits(e.tables[map.prime[r]].p, e.logn))
                return false;
        return true;
    }

    bool ntt_can_gather(const Engine &e)
    {
        return e.use_half_kernel && e.logn >= 14 && e.logn <= 16;
    }

    hipError_t launch_ntt_gather(const Engine &e, u64 *data, size_t nrows, const RowMap &map, const NttSource &src,
                                 int flags)
    {
        if (!ntt_can_gather(e))
            return hipErrorInvalidValue;
        if (e.mode_strict)
            flags |= kNttStrict;
        if (nrows == 0)
            return hipSuccess;
        if (e.logn == 14)
            return launch_half<14>(e, data, nrows, map, flags, src);
        if (e.logn == 15)
            return launch_half<15>(e, data, nrows, map, flags, src);
        return launch_half<16>(e, data, nrows, map, flags, src);
    }

    // inverse NTT of rows read from another buffer (polynomial stride src_poly_stride words), written to data
    hipError_t launch_intt_from(const Engine &e, u64 *data, const u64 *src, std::size_t src_poly_stride, size_t nrows,
                                const RowMap &map, int flags)
    {
        if (!(e.use_half_kernel && e.logn >= 14 && e.logn <= 16))
            return hipErrorInvalidValue;
        if (nrows == 0)
            return hipSuccess;
        if (e.logn == 14)
            return launch_half_inv<14>(e, data, nrows, map, flags, src, src_poly_stride);
        if (e.logn == 15)
            return launch_half_inv<15>(e, data, nrows, map, flags, src, src_poly_stride);
        return launch_half_inv<16>(e, data, nrows, map, flags, src, src_poly_stride);
    }

    // inverse NTT of the ciphertext tensor product of two size-2 operands, formed on load from the forward-transformed
    // rows x (item-major: 4 polynomials of kb rows each, item_stride words apart); map has 3 * kb rows (output polynomial
    // I, row r at slot I * kb + r); the stored values carry the Montgomery factor 2^-64 (see DyadicSrc)
    hipError_t launch_intt_tensor(const Engine &e, u64 *data, const u64 *x, std::size_t item_stride, std::size_t poly_stride,
                                  int kb, size_t nrows, const RowMap &map, int flags, bool square)
    {
        if (!(e.use_half_kernel && e.logn >= 14 && e.logn <= 16) || map.rows != 3 * kb)
            return hipErrorInvalidValue;
        if (nrows == 0)
            return hipSuccess;
        const DyadicSrc dy{ x, item_stride, poly_stride, kb, square ? 1 : 0 };
        if (e.logn == 14)
            return launch_half_inv<14>(e, data, nrows, map, flags, nullptr, 0, &dy);
        if (e.logn == 15)
            return launch_half_inv<15>(e, data, nrows, map, flags, nullptr, 0, &dy);
        return launch_half_inv<16>(e, data, nrows, map, flags, nullptr, 0, &dy);
    }

    hipError_t launch_ntt(const Engine &e, u64 *data, size_t nrows, const RowMap &map, bool inverse, int flags)
    {
        if (e.mode_strict)
            flags |= kNttStrict;
        if (!inverse && e.use_half_kernel && nrows > 0)
        {
            // single-pass forward transform for the large rings
            NttSource none{};
            if (e.logn == 14)
                return launch_half<14>(e, data, nrows, map, flags, none);
            if (e.logn == 15)
                return launch_half<15>(e, data, nrows, map, flags, none);
            if (e.logn == 16)
                return launch_half<16>(e, data, nrows, map, flags, none);
        }
        if (inverse && e.use_half_kernel && nrows > 0)
        {
            if (e.logn == 14)
                return launch_half_inv<14>(e, data, nrows, map, flags);
            if (e.logn == 15)
                return launch_half_inv<15>(e, data, nrows, map, flags);
            if (e.logn == 16)
                return launch_half_inv<16>(e, data, nrows, map, flags);
        }
        const NttPlan plan = plan_ntt(e.logn, inverse, flags);
        return inverse ? launch_dir<1>(e, data, nrows, map, plan) : launch_dir<0>(e, data, nrows, map, plan);
    }
} // namespace sealhip
