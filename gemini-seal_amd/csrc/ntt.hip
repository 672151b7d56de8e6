// ntt.hip -- batched negacyclic NTT / inverse NTT for gfx950 (CDNA4).
//
// Replaces util::ntt_negacyclic_harvey{,_lazy} / inverse_ntt_negacyclic_harvey{,_lazy}
// (native/src/seal/util/ntt.cpp:292-404, ntt.h:225-334) for batches of RNS rows.
//
// Shape of the computation: integer, HBM-streaming, ALU-heavy (one Shoup butterfly = one
// 64x64->hi64 and two 64x64->lo64 products built from v_mad_u64_u32); no MFMA.
//
// Decomposition: a row of N = 2^logn coefficients is transformed in one pass (logn <= 13) or two
// passes. A pass gives each workgroup a TILE of 2^t coefficients that is closed under the
// butterflies of the pass: the tile is staged HBM -> LDS with 16-byte coalesced loads, then the
// threads run register-resident radix-16 rounds (16 coefficients = 4 index bits per thread, up to
// 4 butterfly layers per LDS round trip), and the tile is stored back with 16-byte coalesced
// stores. The two-pass split for logn > 13 is the classic "strided columns, then contiguous rows":
//   strided pass   : tile = {top active bits} x {2^c contiguous coefficients}, global bits [b_lo, logn)
//   contiguous pass: tile = 2^t contiguous coefficients, global bits [0, b_lo)
// Bit-exactness: every butterfly is exactly the reference's radix-2 lazy butterfly,
// only the schedule differs, so the 64-bit words (including the wrap-around behaviour for 60-bit
// primes) are identical. Twiddle of the butterfly on global bit b whose lower element
// has index j: table entry (N + j) >> (b + 1), in both directions.
//
// LDS image: index l is stored at l + (l >> 4) (one pad word per 16), which makes both the
// stride-1 and the stride-16 access patterns of the rounds bank-conflict free for ds_read_b64.
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
#pragma unroll
                        for (int s = 0; s < 16; s++)
                        {
                            if (s & bit)
                                continue;
                            const ulonglong2 W = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (tb + (s >> (w + 1))));
                            u64 u = x[s];
                            if (strict)
                                u = u >= two_p ? u - two_p : u;
                            else if (last)
                                u = barrett_lazy(u, P.rdp, p); // ForwardLazyLast, ntt.cpp:254-261
                            const u64 v = mulmod_lazy(x[s | bit], W.x, W.y, p);
                            x[s] = u + v;               // ForwardLazy, ntt.cpp:245-252
                            x[s | bit] = u - v + two_p;
                        }
                    }
                    else
                    {
                        const bool top = gb == logn - 1;
#pragma unroll
                        for (int s = 0; s < 16; s++)
                        {
                            if (s & bit)
                                continue;
                            ulonglong2 W;
                            if (top)
                            {
                                W.x = P.inv_n_w;
                                W.y = P.inv_n_w_shoup;
                            }
                            else
                                W = *reinterpret_cast<const ulonglong2 *>(tw + 2 * (tb + (s >> (w + 1))));
                            const u64 u = x[s], v = x[s | bit];
                            u64 tt = u + v;
                            tt = tt >= two_p ? tt - two_p : tt; // BackwardLazy, ntt.cpp:265-272
                            if (top)
                                tt = mulmod_lazy(tt, P.inv_n, P.inv_n_shoup, p); // BackwardLazyLast, :274-281
                            x[s] = tt;
                            x[s | bit] = mulmod_lazy(u - v + two_p, W.x, W.y, p);
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
        if (plan.serial)
        {
            plan.npass = 0;
            plan.flags = flags;
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
        if (logn <= kTileBitsMax)
        {
            plan.npass = 1;
            init(plan.pass[0], logn, 0, 0, 0, logn - 1);
        }
        else
        {
            const int cbits = (logn + 1) / 2; // contiguous pass: global bits [0, cbits)
            const int sbits = logn - cbits;   // strided pass:    global bits [cbits, logn)
            const int t = kTileBitsMax;
            NttPass strided{}, contiguous{};
            init(strided, t, t - sbits, cbits, t - sbits, t - 1);
            init(contiguous, t, 0, 0, 0, cbits - 1);
            plan.npass = 2;
            plan.pass[0] = inverse ? contiguous : strided;
            plan.pass[1] = inverse ? strided : contiguous;
        }
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
            ntt_serial_kernel<DIR><<<blocks, threads, 0, e.stream>>>(data, e.d_primes, map, plan.logn, plan.flags, nrows);
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
                ProfScope prof(e, DIR == 0 ? "ntt_fwd_pass" : "ntt_inv_pass", static_cast<double>(nrows));
                ntt_pass_kernel<DIR><<<static_cast<unsigned>(blocks), threads, lds_bytes, e.stream>>>(data, e.d_primes,
                                                                                                      map, ps);
                err = hipGetLastError();
            }
            if (err != hipSuccess)
                return err;
        }
        return hipSuccess;
    }

    hipError_t ntt_init_kernels()
    {
        const int max_lds = ((1 << kTileBitsMax) + (1 << (kTileBitsMax - 4))) * 8;
        hipError_t err = hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<0>),
                                             hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
        if (err != hipSuccess)
            return err;
        return hipFuncSetAttribute(reinterpret_cast<const void *>(&ntt_pass_kernel<1>),
                                   hipFuncAttributeMaxDynamicSharedMemorySize, max_lds);
    }

    hipError_t launch_ntt(const Engine &e, u64 *data, size_t nrows, const RowMap &map, bool inverse, int flags)
    {
        if (e.mode_strict)
            flags |= kNttStrict;
        const NttPlan plan = plan_ntt(e.logn, inverse, flags);
        return inverse ? launch_dir<1>(e, data, nrows, map, plan) : launch_dir<0>(e, data, nrows, map, plan);
    }
} // namespace sealhip
