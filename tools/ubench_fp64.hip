// ubench_fp64.hip -- would double-precision FMA butterflies beat the integer Shoup butterflies for primes below 2^50?
// A lazy modular product in FP64 (h = y*w; l = fma(y,w,-h); q = rint(h/p); r = fma(-q,p,h) + l) is exact integer arithmetic
// as long as every value stays below 2^53 in magnitude: 5 FP64 instructions instead of the 10 multiplier + 6 other integer
// instructions of devmath.hpp. This measures the FMA rate and the butterfly rate on the device (the kernels run at the
// package power cap, so instruction counts alone do not decide it) and checks the arithmetic against __int128 on the host.
// Build: hipcc --offload-arch=gfx950 -O3 -ffp-contract=off tools/ubench_fp64.hip -o tools/bin/ubench_fp64
#include <hip/hip_runtime.h>

#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <vector>

typedef unsigned long long u64;
#define CK(x)                                                                                  \
    do                                                                                         \
    {                                                                                          \
        hipError_t e_ = (x);                                                                   \
        if (e_ != hipSuccess)                                                                  \
        {                                                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);      \
            exit(1);                                                                           \
        }                                                                                      \
    } while (0)
constexpr int ITERS = 2048;

__global__ void k_fma(double *out, double a0, double b0)
{
    double a = a0 + threadIdx.x * 1e-9, b = b0 + blockIdx.x * 1e-9;
    double x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < ITERS; i++)
    {
        x0 = __builtin_fma(x0, a, x1);
        x1 = __builtin_fma(x1, b, x2);
        x2 = __builtin_fma(x2, a, x3);
        x3 = __builtin_fma(x3, b, x0);
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 + x1 + x2 + x3;
}

// signed lazy butterfly: X = u + r, Y = u - r with r = y*w - rint(y*w/p)*p exactly (|r| <= ~1.3p for |y| < 2^51)
__device__ __forceinline__ void bf(double &u, double &y, double w, double p, double pinv)
{
    const double h = y * w;
    const double l = __builtin_fma(y, w, -h);
    const double q = __builtin_rint(h * pinv);
    const double r = __builtin_fma(-q, p, h) + l;
    y = u - r;
    u = u + r;
}
__device__ __forceinline__ double red(double x, double p, double pinv)
{
    return __builtin_fma(-__builtin_rint(x * pinv), p, x);
}

// 4 independent pairs per lane, per-lane twiddles, a reduction of every value each third layer (the bound schedule of a
// 50-bit prime: values below 8p = 2^53)
__global__ void k_butterfly_fp(double *out, double p, double pinv, u64 seed)
{
    const unsigned tid = blockIdx.x * blockDim.x + threadIdx.x;
    double u[4], y[4], w[4];
    for (int j = 0; j < 4; j++)
    {
        u[j] = (double)((seed * (tid * 8 + 2 * j + 1)) % (u64)p);
        y[j] = (double)((seed * (tid * 8 + 2 * j + 2) * 0x9E3779B97F4A7C15ull) % (u64)p);
        w[j] = (double)((seed + tid * 4 + j) * 0xD1B54A32D192ED03ull % (u64)p);
    }
    for (int i = 0; i < ITERS; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
            bf(u[j], y[j], w[j], p, pinv);
        if (i % 3 == 2)
        {
#pragma unroll
            for (int j = 0; j < 4; j++)
            {
                u[j] = red(u[j], p, pinv);
                y[j] = red(y[j], p, pinv);
            }
        }
    }
    double acc = 0;
    for (int j = 0; j < 4; j++)
        acc += red(u[j], p, pinv) + red(y[j], p, pinv);
    out[tid] = acc;
}

// exactness check: one butterfly chain of `steps` layers on the device vs __int128 on the host
__global__ void k_check(const u64 *in_u, const u64 *in_y, const u64 *in_w, long long *out_u, long long *out_y, double p,
                        double pinv, int steps, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n)
        return;
    double u = (double)in_u[i], y = (double)in_y[i];
    const double w = (double)in_w[i];
    for (int s = 0; s < steps; s++)
    {
        bf(u, y, w, p, pinv);
        if (s % 3 == 2)
        {
            u = red(u, p, pinv);
            y = red(y, p, pinv);
        }
    }
    out_u[i] = (long long)red(u, p, pinv);
    out_y[i] = (long long)red(y, p, pinv);
}

template <class F>
double time_ms(F f)
{
    f();
    CK(hipDeviceSynchronize());
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++)
        f();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

int main()
{
    const int blocks = 256 * 8, threads = 256;
    double *out;
    CK(hipMalloc(&out, sizeof(double) * blocks * threads));
    const double lanes = (double)blocks * threads;
    const u64 pi = 1125899903107073ull; // a 50-bit NTT prime of config 2
    const double p = (double)pi, pinv = 1.0 / p;
    double ms = time_ms([&] { k_fma<<<blocks, threads>>>(out, 1.0000001, 0.9999999); });
    printf("v_fma_f64           : %8.2f G lane-ops/s (%.3f ms)\n", lanes * ITERS * 4 / ms / 1e6, ms);
    ms = time_ms([&] { k_butterfly_fp<<<blocks, threads>>>(out, p, pinv, 12345); });
    printf("fp64 lazy butterfly : %8.2f G butterflies/s incl. a reduction of every value each third layer (%.3f ms)\n",
           lanes * ITERS * 4 / ms / 1e6, ms);
    for (int t = 256; t <= 1024; t *= 2)
    {
        double a0 = time_ms([&] { k_butterfly_fp<<<256, t>>>(out, p, pinv, 999); });
        printf("  %d wave(s)/SIMD: %8.2f G butterflies/s\n", t / 256, 256.0 * t * ITERS * 4 / a0 / 1e6);
    }
    // exactness against __int128
    const int n = 1 << 16, steps = 15;
    std::vector<u64> hu(n), hy(n), hw(n);
    u64 s = 88172645463325252ull;
    auto rnd = [&] {
        s ^= s << 13;
        s ^= s >> 7;
        s ^= s << 17;
        return s;
    };
    for (int i = 0; i < n; i++)
    {
        hu[i] = rnd() % pi;
        hy[i] = rnd() % pi;
        hw[i] = i < 16 ? pi - 1 : rnd() % pi;
        if (i < 8)
            hu[i] = hy[i] = pi - 1;
    }
    u64 *du, *dy, *dw;
    long long *ou, *oy;
    CK(hipMalloc(&du, 8 * n));
    CK(hipMalloc(&dy, 8 * n));
    CK(hipMalloc(&dw, 8 * n));
    CK(hipMalloc(&ou, 8 * n));
    CK(hipMalloc(&oy, 8 * n));
    CK(hipMemcpy(du, hu.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dy, hy.data(), 8 * n, hipMemcpyHostToDevice));
    CK(hipMemcpy(dw, hw.data(), 8 * n, hipMemcpyHostToDevice));
    k_check<<<n / 256, 256>>>(du, dy, dw, ou, oy, p, pinv, steps, n);
    std::vector<long long> gu(n), gy(n);
    CK(hipMemcpy(gu.data(), ou, 8 * n, hipMemcpyDeviceToHost));
    CK(hipMemcpy(gy.data(), oy, 8 * n, hipMemcpyDeviceToHost));
    size_t bad = 0;
    for (int i = 0; i < n; i++)
    {
        unsigned __int128 u = hu[i], y = hy[i];
        for (int st = 0; st < steps; st++)
        {
            const unsigned __int128 v = y * hw[i] % pi;
            const unsigned __int128 nu = (u + v) % pi, ny = (u + pi - v) % pi;
            u = nu;
            y = ny;
        }
        const long long a = ((gu[i] % (long long)pi) + (long long)pi) % (long long)pi;
        const long long b = ((gy[i] % (long long)pi) + (long long)pi) % (long long)pi;
        bad += (u64)a != (u64)u || (u64)b != (u64)y;
    }
    printf("15 butterfly layers in FP64 vs __int128 on %d lanes: %zu mismatches\n", n, bad);
    return bad != 0;
}
