// ckks_encoder.hip -- CKKSEncoder::encode / decode (native/src/seal/ckks.cpp:14-77, ckks.h:405-747) on the device
// (SURVEY 8 f4). The only floating-point part of the engine: double-precision special FFTs over the 2N-th complex roots.
// Every floating-point operation is issued in the order the reference's std::complex<double> arithmetic performs it and
// contraction into FMAs is switched off for this file, so the results are the reference's bits (the root tables come
// from the host's libm, hostmath.cpp). The integer side (rounding, RNS decomposition, CRT composition) is exact.
#include "engine.hpp"

#pragma clang fp contract(off)

namespace sealhip
{
    namespace
    {
        constexpr int kThreads = 256;
        constexpr int kTileLog = 11; // 2048 complex numbers = 32 KiB of LDS per workgroup

        inline unsigned grid_for(std::size_t work_items)
        {
            std::size_t blocks = (work_items + kThreads - 1) / kThreads;
            const std::size_t cap = 256u * 32u;
            return static_cast<unsigned>(blocks < cap ? (blocks ? blocks : 1) : cap);
        }

        __device__ __forceinline__ double2 cmul(double2 a, double2 b) // std::complex operator* (no NaN recovery needed)
        {
            const double ac = a.x * b.x, bd = a.y * b.y, ad = a.x * b.y, bc = a.y * b.x;
            return make_double2(ac - bd, ad + bc);
        }

        // ckks.h:451-456 as a gather through the inverse slot table (stored behind the table): conj_values[p] =
        // values[j] for p = map[j], conj(values[j]) for p = map[j + slots], zero for slots beyond n_values
        __global__ __launch_bounds__(kThreads) void ckks_place_kernel(const double2 *__restrict__ values,
                                                                      std::size_t n_values, double2 *__restrict__ cv,
                                                                      const std::uint32_t *__restrict__ map, int logn,
                                                                      std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t n = std::size_t(1) << logn, slots = n >> 1;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> logn, p = i & (n - 1);
                const std::uint32_t j = map[n + p];
                const std::size_t slot = j < slots ? j : j - slots;
                double2 v = make_double2(0.0, 0.0);
                if (slot < n_values)
                {
                    v = values[item * n_values + slot];
                    if (j >= slots)
                        v.y = -v.y;
                }
                cv[i] = v;
            }
        }

        // one butterfly of the inverse (encode, ckks.h:471-477) or forward (decode, :733-739) special FFT
        template <bool INV>
        __device__ __forceinline__ void bfly(double2 &a, double2 &b, double2 s)
        {
            const double2 u = a;
            if (INV)
            {
                const double2 v = b;
                a = make_double2(u.x + v.x, u.y + v.y);
                b = cmul(make_double2(u.x - v.x, u.y - v.y), s);
            }
            else
            {
                const double2 v = cmul(b, s);
                a = make_double2(u.x + v.x, u.y + v.y);
                b = make_double2(u.x - v.x, u.y - v.y);
            }
        }

        // one layer in global memory: one lane per butterfly. INV layer i pairs k and k + 2^i (twiddle inv_roots[h + j],
        // h = n >> (i+1)); forward layer i pairs k and k + (n >> (i+1)) (twiddle roots[2^i + j]).
        template <bool INV>
        __global__ __launch_bounds__(kThreads) void fft_layer_kernel(double2 *__restrict__ data,
                                                                     const double2 *__restrict__ roots, int logn, int layer,
                                                                     std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t half = std::size_t(1) << (logn - 1);
            const int sh = INV ? layer : logn - layer - 1; // log2 of the pair distance tt
            const std::size_t tt = std::size_t(1) << sh;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> (logn - 1), b = i & (half - 1);
                const std::size_t j = b >> sh, kk = b & (tt - 1);
                const std::size_t k = (j << (sh + 1)) + kk;
                const double2 s = roots[(INV ? (half >> layer) : (std::size_t(1) << layer)) + j];
                double2 *p = data + (item << logn) + k;
                double2 x = p[0], y = p[tt];
                bfly<INV>(x, y, s);
                p[0] = x;
                p[tt] = y;
            }
        }

        // two layers in global memory per launch: one lane per quadruple. Every element goes through the same two
        // butterflies with the same operands as in two single-layer launches, so the bits are unchanged.
        // INV layers (i, i+1), tt = 2^i: elements k, k+tt, k+2tt, k+3tt; FWD layers (i, i+1), tt = n >> (i+1):
        // elements k, k+tt/2, k+tt, k+tt+tt/2.
        template <bool INV>
        __global__ __launch_bounds__(kThreads) void fft_layer2_kernel(double2 *__restrict__ data,
                                                                      const double2 *__restrict__ roots, int logn, int layer,
                                                                      std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t half = std::size_t(1) << (logn - 1), quads = half >> 1;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> (logn - 2), q = i & (quads - 1);
                double2 *base = data + (item << logn);
                if (INV)
                {
                    const int sh = layer;
                    const std::size_t tt = std::size_t(1) << sh;
                    const std::size_t g = q >> sh, kk = q & (tt - 1);
                    double2 *p = base + (g << (sh + 2)) + kk;
                    double2 e0 = p[0], e1 = p[tt], e2 = p[2 * tt], e3 = p[3 * tt];
                    const std::size_t h0 = half >> layer, h1 = half >> (layer + 1);
                    bfly<true>(e0, e1, roots[h0 + 2 * g]);
                    bfly<true>(e2, e3, roots[h0 + 2 * g + 1]);
                    const double2 s2 = roots[h1 + g];
                    bfly<true>(e0, e2, s2);
                    bfly<true>(e1, e3, s2);
                    p[0] = e0;
                    p[tt] = e1;
                    p[2 * tt] = e2;
                    p[3 * tt] = e3;
                }
                else
                {
                    const int sh = logn - layer - 2; // log2 of tt/2
                    const std::size_t t2 = std::size_t(1) << sh;
                    const std::size_t j = q >> sh, kk = q & (t2 - 1);
                    double2 *p = base + (j << (sh + 2)) + kk;
                    double2 e0 = p[0], e1 = p[t2], e2 = p[2 * t2], e3 = p[3 * t2];
                    const std::size_t m0 = std::size_t(1) << layer, m1 = m0 << 1;
                    const double2 s0 = roots[m0 + j];
                    bfly<false>(e0, e2, s0);
                    bfly<false>(e1, e3, s0);
                    bfly<false>(e0, e1, roots[m1 + 2 * j]);
                    bfly<false>(e2, e3, roots[m1 + 2 * j + 1]);
                    p[0] = e0;
                    p[t2] = e1;
                    p[2 * t2] = e2;
                    p[3 * t2] = e3;
                }
            }
        }

        // L consecutive layers whose pairs stay inside a contiguous tile of 2^L numbers, in LDS: the first L layers of
        // the inverse transform, the last L of the forward one.
        template <bool INV>
        __global__ __launch_bounds__(kThreads) void fft_local_kernel(double2 *__restrict__ data,
                                                                     const double2 *__restrict__ roots, int logn, int L)
        {
            extern __shared__ double2 tile[];
            const std::size_t T = std::size_t(1) << L;
            const std::size_t tiles_per_item = std::size_t(1) << (logn - L);
            const std::size_t base = (blockIdx.x & (tiles_per_item - 1)) << L; // element index inside the item
            double2 *g = data + static_cast<std::size_t>(blockIdx.x) * T;
            for (std::size_t i = threadIdx.x; i < T; i += kThreads)
                tile[i] = g[i];
            __syncthreads();
            const std::size_t half = std::size_t(1) << (logn - 1);
            for (int step = 0; step < L; step++)
            {
                const int layer = INV ? step : logn - L + step;
                const int sh = INV ? layer : logn - layer - 1;
                const std::size_t tt = std::size_t(1) << sh;
                for (std::size_t bl = threadIdx.x; bl < T / 2; bl += kThreads)
                {
                    const std::size_t gl = bl >> sh, kk = bl & (tt - 1);
                    const std::size_t k = (gl << (sh + 1)) + kk;
                    const std::size_t j = (base >> (sh + 1)) + gl;
                    const double2 s = roots[(INV ? (half >> layer) : (std::size_t(1) << layer)) + j];
                    bfly<INV>(tile[k], tile[k + tt], s);
                }
                __syncthreads();
            }
            for (std::size_t i = threadIdx.x; i < T; i += kThreads)
                g[i] = tile[i];
        }

        // ckks.h:484-607: scale, record the largest bit count, round to nearest (ties away from zero), reduce the exact
        // integer modulo every prime and put the sign back. A rounded double is m * 2^e with a 53-bit m, so the
        // multi-precision decomposition of the slow path (:569-607) is m * (2^64)^(e/64) * 2^(e%64) mod q_j -- the same
        // canonical residue every one of the reference's three paths yields.
        __global__ __launch_bounds__(kThreads) void ckks_round_decompose_kernel(const double2 *__restrict__ cv, double n_inv,
                                                                                u64 *__restrict__ out, int rows,
                                                                                const PrimeDev *__restrict__ primes, int logn,
                                                                                int *__restrict__ max_bits, std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t n = std::size_t(1) << logn;
            int local_max = 1;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> logn, c = i & (n - 1);
                const double x = cv[i].x * n_inv;
                const double d = fmax(fabs(x), 1.0);
                const int bits = ilogb(d) + 2; // static_cast<int>(log2(d)) + 2, :498-499
                local_max = bits > local_max ? bits : local_max;
                const double r = round(x);
                const bool negative = signbit(r);
                const double a = fabs(r);
                int e2 = 0;
                u64 mant;
                if (a < 9007199254740992.0) // 2^53: the integer itself
                    mant = static_cast<u64>(a);
                else
                {
                    int ex;
                    const double f = frexp(a, &ex); // a = f * 2^ex, f in [0.5, 1)
                    mant = static_cast<u64>(ldexp(f, 53));
                    e2 = ex - 53;
                }
                const int limb = e2 >> 6, bit = e2 & 63;
                // the two 64-bit limbs the mantissa occupies
                const u64 lo = mant << bit, hi = bit ? mant >> (64 - bit) : 0;
                u64 *dst = out + item * static_cast<std::size_t>(rows) * n + c;
                for (int j = 0; j < rows; j++)
                {
                    const PrimeDev &P = primes[j];
                    u64 v = barrett_reduce_128(lo, hi, P.p, P.cr0, P.cr1);
                    if (limb)
                    {
                        const u64 w = barrett_reduce_128(0, 1, P.p, P.cr0, P.cr1); // 2^64 mod p
                        for (int l = 0; l < limb; l++)
                            v = mul_mod(v, w, P.p, P.cr0, P.cr1);
                    }
                    dst[static_cast<std::size_t>(j) << logn] = negative ? neg_mod(v, P.p) : v;
                }
            }
            if (local_max > 1)
                atomicMax(max_bits, local_max);
        }

        // RNSBase::compose_array (rns.cpp:401-450) + ckks.h:681-720: CRT-compose a coefficient into K limbs, compare
        // with the upper half threshold, and sum the limbs into a double in the reference's order.
        template <int KMAX>
        __global__ __launch_bounds__(kThreads) void ckks_compose_kernel(const u64 *__restrict__ coeff, const CkksDecodeDev *d_,
                                                                        const PrimeDev *__restrict__ primes, double inv_scale,
                                                                        double2 *__restrict__ res, int logn, std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t n = std::size_t(1) << logn;
            const CkksDecodeDev &d = *d_;
            const int K = d.k;
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> logn, c = i & (n - 1);
                const u64 *src = coeff + item * static_cast<std::size_t>(K) * n + c;
                u64 acc[KMAX + 1];
#pragma unroll
                for (int l = 0; l <= KMAX; l++)
                    acc[l] = 0;
                for (int r = 0; r < K; r++)
                {
                    const PrimeDev &P = primes[r];
                    const u64 t = mul_mod(src[static_cast<std::size_t>(r) << logn], d.inv_punct[r], P.p, P.cr0, P.cr1);
                    const u64 *pp = d.punct + r * kCkksMaxLimbs;
                    u64 carry = 0;
#pragma unroll
                    for (int l = 0; l < KMAX; l++)
                        if (l < K)
                        {
                            // acc[l] + t * pp[l] + carry  (< 2^128)
                            u64 lo = acc[l], hi = 0;
                            mac128(lo, hi, t, pp[l]);
                            const u64 s = lo + carry;
                            hi += s < lo;
                            acc[l] = s;
                            carry = hi;
                        }
                    u64 top = carry; // limb K of the sum (acc[K] is always zero between rounds)
                    // one conditional subtraction of Q (the sum is below 2Q)
                    bool ge = top != 0;
                    if (!ge)
                    {
                        ge = true;
                        bool decided = false;
#pragma unroll
                        for (int l = KMAX - 1; l >= 0; l--)
                            if (l < K && !decided && acc[l] != d.q[l])
                            {
                                ge = acc[l] > d.q[l];
                                decided = true;
                            }
                    }
                    if (ge)
                    {
                        u64 borrow = 0;
#pragma unroll
                        for (int l = 0; l < KMAX; l++)
                            if (l < K)
                            {
                                const u64 a = acc[l], b = d.q[l];
                                const u64 df = a - b - borrow;
                                borrow = (a < b) || (a == b && borrow) ? 1 : 0;
                                acc[l] = df;
                            }
                    }
                }
                bool upper = true, decided = false; // is_greater_than_or_equal_uint(acc, upper_half_threshold)
#pragma unroll
                for (int l = KMAX - 1; l >= 0; l--)
                    if (l < K && !decided && acc[l] != d.half[l])
                    {
                        upper = acc[l] > d.half[l];
                        decided = true;
                    }
                double r = 0.0, scaled = inv_scale;
                const double two_pow_64 = 18446744073709551616.0;
#pragma unroll
                for (int j = 0; j < KMAX; j++)
                    if (j < K)
                    {
                        if (upper)
                        {
                            if (acc[j] > d.q[j])
                            {
                                const u64 diff = acc[j] - d.q[j];
                                r += diff ? static_cast<double>(diff) * scaled : 0.0;
                            }
                            else
                            {
                                const u64 diff = d.q[j] - acc[j];
                                r -= diff ? static_cast<double>(diff) * scaled : 0.0;
                            }
                        }
                        else
                            r += acc[j] ? static_cast<double>(acc[j]) * scaled : 0.0;
                        scaled *= two_pow_64;
                    }
                res[i] = make_double2(r, 0.0);
            }
        }

        // ckks.h:743-746: destination[i] = res[map[i]]
        __global__ __launch_bounds__(kThreads) void ckks_pick_kernel(const double2 *__restrict__ res, double2 *__restrict__ values,
                                                                     const std::uint32_t *__restrict__ map, int logn,
                                                                     std::size_t total)
        {
            const std::size_t stride = static_cast<std::size_t>(gridDim.x) * blockDim.x;
            const std::size_t slots = std::size_t(1) << (logn - 1);
            for (std::size_t i = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x; i < total; i += stride)
            {
                const std::size_t item = i >> (logn - 1), s = i & (slots - 1);
                values[i] = res[(item << logn) + map[s]];
            }
        }

        template <bool INV>
        hipError_t run_fft(const Engine &e, double2 *data, const double2 *roots, std::size_t count)
        {
            const int logn = e.logn, L = logn < kTileLog ? logn : kTileLog;
            const std::size_t tiles = count << (logn - L);
            const std::size_t lds = sizeof(double2) << L;
            const std::size_t bflies = count << (logn - 1);
            const std::size_t quads = bflies / 2;
            if (INV)
            {
                fft_local_kernel<true><<<static_cast<unsigned>(tiles), kThreads, lds, e.lane().stream>>>(data, roots, logn, L);
                int layer = L;
                for (; layer + 1 < logn; layer += 2)
                    fft_layer2_kernel<true><<<grid_for(quads), kThreads, 0, e.lane().stream>>>(data, roots, logn, layer, quads);
                if (layer < logn)
                    fft_layer_kernel<true><<<grid_for(bflies), kThreads, 0, e.lane().stream>>>(data, roots, logn, layer, bflies);
            }
            else
            {
                int layer = 0;
                for (; layer + 1 < logn - L; layer += 2)
                    fft_layer2_kernel<false><<<grid_for(quads), kThreads, 0, e.lane().stream>>>(data, roots, logn, layer, quads);
                if (layer < logn - L)
                    fft_layer_kernel<false><<<grid_for(bflies), kThreads, 0, e.lane().stream>>>(data, roots, logn, layer, bflies);
                fft_local_kernel<false><<<static_cast<unsigned>(tiles), kThreads, lds, e.lane().stream>>>(data, roots, logn, L);
            }
            return hipGetLastError();
        }
    } // namespace

    hipError_t launch_ckks_encode_front(const Engine &e, const double *values, std::size_t n_values, std::size_t count,
                                        double n_inv_scale, double *cv, u64 *out, int rows, const std::uint32_t *map,
                                        const double *inv_roots, int *max_bits)
    {
        const std::size_t total = count << e.logn;
        if (!total)
            return hipSuccess;
        ProfScope prof(e, "ckks_encode_fft", static_cast<double>(total));
        double2 *c2 = reinterpret_cast<double2 *>(cv);
        ckks_place_kernel<<<grid_for(total), kThreads, 0, e.lane().stream>>>(reinterpret_cast<const double2 *>(values), n_values, c2, map,
                                                                      e.logn, total);
        hipError_t err = run_fft<true>(e, c2, reinterpret_cast<const double2 *>(inv_roots), count);
        if (err != hipSuccess)
            return err;
        ckks_round_decompose_kernel<<<grid_for(total), kThreads, 0, e.lane().stream>>>(c2, n_inv_scale, out, rows, e.d_primes, e.logn,
                                                                                max_bits, total);
        return hipGetLastError();
    }

    hipError_t launch_ckks_decode_back(const Engine &e, const u64 *coeff, const CkksDecodeDev *d, int k, std::size_t count,
                                       double inv_scale, double *res, double *values, const std::uint32_t *map,
                                       const double *roots)
    {
        const std::size_t total = count << e.logn;
        if (!total)
            return hipSuccess;
        ProfScope prof(e, "ckks_decode_fft", static_cast<double>(total));
        double2 *r2 = reinterpret_cast<double2 *>(res);
        if (k <= 4)
            ckks_compose_kernel<4><<<grid_for(total), kThreads, 0, e.lane().stream>>>(coeff, d, e.d_primes, inv_scale, r2, e.logn, total);
        else if (k <= 8)
            ckks_compose_kernel<8><<<grid_for(total), kThreads, 0, e.lane().stream>>>(coeff, d, e.d_primes, inv_scale, r2, e.logn, total);
        else if (k <= 16)
            ckks_compose_kernel<16><<<grid_for(total), kThreads, 0, e.lane().stream>>>(coeff, d, e.d_primes, inv_scale, r2, e.logn, total);
        else
            ckks_compose_kernel<kCkksMaxLimbs><<<grid_for(total), kThreads, 0, e.lane().stream>>>(coeff, d, e.d_primes, inv_scale, r2,
                                                                                          e.logn, total);
        hipError_t err = hipGetLastError();
        if (err != hipSuccess)
            return err;
        err = run_fft<false>(e, r2, reinterpret_cast<const double2 *>(roots), count);
        if (err != hipSuccess)
            return err;
        ckks_pick_kernel<<<grid_for(total / 2), kThreads, 0, e.lane().stream>>>(r2, reinterpret_cast<double2 *>(values), map, e.logn,
                                                                        total / 2);
        return hipGetLastError();
    }
} // namespace sealhip
