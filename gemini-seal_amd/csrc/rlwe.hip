// rlwe.hip -- the arithmetic either side of the evaluator path, streaming kernels:
//   * encrypt_zero_symmetric / encrypt_zero_asymmetric with the random samples handed in
//     (native/src/seal/util/rlwe.cpp:140-300; sampling stays on the host),
//   * multiply_add/sub_plain_with_scaling_variant (native/src/seal/util/scalingvariant.cpp:15-92),
//   * the slot permutation of BatchEncoder (native/src/seal/batchencoder.cpp:70-154, :339-376).
// The transforms between the stages are the engine's NTT launches (pipeline.cpp).
#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        constexpr int kThreads = 256;

        inline unsigned grid_for(std::size_t work_items)
        {
            std::size_t blocks = (work_items + kThreads - 1) / kThreads;
            const std::size_t cap = 256u * 16u; // grid-stride the rest
            return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
        }

        // the residue sample_poly_ternary / sample_poly_normal store for a small signed value (rlwe.cpp:25-95)
        __device__ __forceinline__ u64 lift_small(int v, u64 p)
        {
            return v >= 0 ? static_cast<u64>(v) : p - static_cast<u64>(-static_cast<long long>(v));
        }

        // One lane per coefficient pair of (item, poly j < polys, row r < rows). Row r uses prime id r.
        //   STAGE 0: ct = lift(e)                         STAGE 1: ct = x (.) y
        //   STAGE 2: ct = [-] (ct + x (.) y)              STAGE 3: ct = [-] (lift(e) + ct)
        // x: per item (and per poly when x_poly_stride != 0), y: shared by all items, e: small signed samples.
        template <int STAGE>
        __global__ __launch_bounds__(kThreads) void rlwe_stage_kernel(RlweArgs a, const PrimeDev *__restrict__ primes,
                                                                      int logn, std::size_t total_pairs)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t pairs_per_row = std::size_t(1) << (logn - 1);
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total_pairs;
                 i += stride)
            {
                const std::size_t c = (i & (pairs_per_row - 1)) * 2;
                std::size_t rest = i >> (logn - 1);
                const int r = static_cast<int>(rest % a.rows);
                rest /= a.rows;
                const int j = static_cast<int>(rest % a.polys);
                const std::size_t item = rest / a.polys;
                const PrimeDev &P = primes[r];
                const std::size_t row_off = (static_cast<std::size_t>(r) << logn) + c;
                u64 *dst = a.ct + item * a.ct_item_stride + j * a.ct_poly_stride + row_off;
                ulonglong2 out;
                if (STAGE == 0 || STAGE == 3)
                {
                    const int2 ev = *reinterpret_cast<const int2 *>(a.e + item * a.e_item_stride + j * a.e_poly_stride + c);
                    out.x = lift_small(ev.x, P.p);
                    out.y = lift_small(ev.y, P.p);
                    if (STAGE == 3)
                    {
                        const ulonglong2 cur = *reinterpret_cast<const ulonglong2 *>(dst);
                        out.x = add_mod(out.x, cur.x, P.p);
                        out.y = add_mod(out.y, cur.y, P.p);
                    }
                }
                else
                {
                    const ulonglong2 xv =
                        *reinterpret_cast<const ulonglong2 *>(a.x + item * a.x_item_stride + j * a.x_poly_stride + row_off);
                    const ulonglong2 yv = *reinterpret_cast<const ulonglong2 *>(a.y + j * a.y_poly_stride + row_off);
                    out.x = mul_mod(xv.x, yv.x, P.p, P.cr0, P.cr1);
                    out.y = mul_mod(xv.y, yv.y, P.p, P.cr0, P.cr1);
                    if (STAGE == 2)
                    {
                        const ulonglong2 cur = *reinterpret_cast<const ulonglong2 *>(dst);
                        out.x = add_mod(cur.x, out.x, P.p);
                        out.y = add_mod(cur.y, out.y, P.p);
                    }
                }
                if ((STAGE == 2 || STAGE == 3) && a.negate)
                {
                    out.x = neg_mod(out.x, P.p);
                    out.y = neg_mod(out.y, P.p);
                }
                *reinterpret_cast<ulonglong2 *>(dst) = out;
            }
        }

        // scalingvariant.cpp:31-51 / :70-90. One lane per (item, coefficient); the k rows of c0 are walked by the lane
        // (fix is computed once per coefficient, as in the reference). floor(numerator / t) comes from the two-word
        // Barrett quotient with one correction (numerator < t^2 + t < 2^123, quotient <= t).
        __global__ __launch_bounds__(kThreads) void scaling_variant_kernel(ScalingArgs a, const PrimeDev *__restrict__ primes,
                                                                           int logn, std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t n = std::size_t(1) << logn;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> logn, c = i & (n - 1);
                const u64 m = a.plain[item * a.plain_item_stride + c];
                u64 lo = 0, hi = 0;
                mac128(lo, hi, m, a.q_mod_t);
                const u64 lo2 = lo + a.threshold;
                hi += lo2 < lo;
                lo = lo2;
                // quotient of barrett_reduce_128 (uintarithsmallmod.h:140-178) kept instead of thrown away
                u64 fix;
                {
                    const u64 carry0 = mulhi(lo, a.t_cr0);
                    const u64 t1lo = lo * a.t_cr1, t1hi = mulhi(lo, a.t_cr1);
                    const u64 tmp1 = t1lo + carry0;
                    const u64 tmp3 = t1hi + (tmp1 < t1lo);
                    const u64 ulo = hi * a.t_cr0, uhi = mulhi(hi, a.t_cr0);
                    const u64 tmp1b = tmp1 + ulo;
                    const u64 carry1 = uhi + (tmp1b < tmp1);
                    fix = hi * a.t_cr1 + tmp3 + carry1;
                    const u64 rem = lo - fix * a.t;
                    fix += rem >= a.t;
                }
                u64 *dst = a.c0 + item * a.c0_item_stride + c;
                for (int j = 0; j < a.k; j++)
                {
                    const PrimeDev &P = primes[j];
                    const u64 scaled = mul_add_mod(a.div[j], m, fix, P.p, P.cr0, P.cr1);
                    const u64 cur = dst[static_cast<std::size_t>(j) << logn];
                    dst[static_cast<std::size_t>(j) << logn] = a.sub ? sub_mod(cur, scaled, P.p) : add_mod(cur, scaled, P.p);
                }
            }
        }

        // BatchEncoder slot permutation: ENCODE plain[map[i]] = i < nvalues ? values[i] : 0 (batchencoder.cpp:142-149), done as a
        // gather through the inverse table stored behind map;
        // DECODE values[i] = plain[map[i]] (:371-375). One lane per (item, slot).
        template <bool ENCODE>
        __global__ __launch_bounds__(kThreads) void batch_permute_kernel(const u64 *__restrict__ in, u64 *__restrict__ out,
                                                                         const std::uint32_t *__restrict__ map, int logn,
                                                                         std::size_t in_item_stride, std::size_t nvalues,
                                                                         std::size_t total, u64 signed_t)
        {
            // signed_t != 0: the int64 overloads (batchencoder.cpp:156-198, :378-420) -- negative values are stored as t + v,
            // slot values above t/2 come back as v - t (two's complement words)
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t n = std::size_t(1) << logn;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> logn, s = i & (n - 1);
                if (ENCODE)
                {
                    const std::uint32_t src = map[n + s]; // inverse permutation: plain[s] = values[map^{-1}(s)]
                    u64 v = src < nvalues ? in[item * in_item_stride + src] : 0;
                    if (signed_t && static_cast<long long>(v) < 0)
                        v += signed_t;
                    out[(item << logn) + s] = v;
                }
                else
                {
                    u64 v = in[item * in_item_stride + map[s]];
                    if (signed_t && v > (signed_t >> 1))
                        v -= signed_t;
                    out[(item << logn) + s] = v;
                }
            }
        }
    } // namespace

    hipError_t launch_rlwe_stage(const Engine &e, int stage, const RlweArgs &a, std::size_t count)
    {
        const std::size_t total = count * a.polys * a.rows * (e.n / 2);
        if (!total)
            return hipSuccess;
        ProfScope prof(e, "rlwe_stage", static_cast<double>(total));
        const unsigned grid = grid_for(total);
        switch (stage)
        {
        case 0:
            rlwe_stage_kernel<0><<<grid, kThreads, 0, e.lane().stream>>>(a, e.d_primes, e.logn, total);
            break;
        case 1:
            rlwe_stage_kernel<1><<<grid, kThreads, 0, e.lane().stream>>>(a, e.d_primes, e.logn, total);
            break;
        case 2:
            rlwe_stage_kernel<2><<<grid, kThreads, 0, e.lane().stream>>>(a, e.d_primes, e.logn, total);
            break;
        default:
            rlwe_stage_kernel<3><<<grid, kThreads, 0, e.lane().stream>>>(a, e.d_primes, e.logn, total);
            break;
        }
        return hipGetLastError();
    }

    hipError_t launch_scaling_variant(const Engine &e, const ScalingArgs &a, std::size_t count)
    {
        const std::size_t total = count * e.n;
        if (!total)
            return hipSuccess;
        ProfScope prof(e, "scaling_variant", static_cast<double>(total));
        scaling_variant_kernel<<<grid_for(total), kThreads, 0, e.lane().stream>>>(a, e.d_primes, e.logn, total);
        return hipGetLastError();
    }

    hipError_t launch_batch_permute(const Engine &e, bool encode, const u64 *in, std::size_t in_item_stride,
                                    std::size_t nvalues, u64 *out, const std::uint32_t *map, std::size_t count, u64 signed_t)
    {
        const std::size_t total = count * e.n;
        if (!total)
            return hipSuccess;
        ProfScope prof(e, "batch_permute", static_cast<double>(total));
        if (encode)
            batch_permute_kernel<true><<<grid_for(total), kThreads, 0, e.lane().stream>>>(in, out, map, e.logn, in_item_stride,
                                                                                  nvalues, total, signed_t);
        else
            batch_permute_kernel<false><<<grid_for(total), kThreads, 0, e.lane().stream>>>(in, out, map, e.logn, in_item_stride,
                                                                                   nvalues, total, signed_t);
        return hipGetLastError();
    }
} // namespace sealhip
