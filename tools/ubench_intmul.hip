// ubench_intmul.hip -- how fast can gfx950 do the integer work of a Shoup butterfly?
// Measures wave-level throughput of v_mad_u64_u32 / 64-bit mulhi / full lazy butterflies, to size the
// NTT against its ALU bound (DESIGN.md "NTT: ALU vs HBM"). Build: hipcc --offload-arch=gfx950 -O3.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
#define CK(x)                                                                  \
    do                                                                         \
    {                                                                          \
        hipError_t e_ = (x);                                                   \
        if (e_ != hipSuccess)                                                  \
        {                                                                      \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            exit(1);                                                           \
        }                                                                      \
    } while (0)

constexpr int ITERS = 4096;

__global__ void k_mad64(u64 *out, unsigned a0, unsigned b0)
{
    unsigned a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    u64 acc0 = threadIdx.x, acc1 = 1, acc2 = 2, acc3 = 3;
    for (int i = 0; i < ITERS; i++)
    {
        acc0 = (u64)a * (unsigned)acc1 + acc0; // v_mad_u64_u32
        acc1 = (u64)b * (unsigned)acc2 + acc1;
        acc2 = (u64)a * (unsigned)acc3 + acc2;
        acc3 = (u64)b * (unsigned)acc0 + acc3;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0 ^ acc1 ^ acc2 ^ acc3;
}

__global__ void k_mullo32(u64 *out, unsigned a0, unsigned b0)
{
    unsigned a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    unsigned x0 = threadIdx.x, x1 = 1, x2 = 2, x3 = 3;
    for (int i = 0; i < ITERS; i++)
    {
        x0 = x0 * a + x1; // v_mul_lo_u32 + add (or v_mad_u32_u24? no: full 32-bit)
        x1 = x1 * b + x2;
        x2 = x2 * a + x3;
        x3 = x3 * b + x0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

__global__ void k_mulhi64(u64 *out, u64 a0, u64 b0)
{
    u64 a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    for (int i = 0; i < ITERS; i++)
    {
        x0 = __umul64hi(x0, a) + x1;
        x1 = __umul64hi(x1, b) + x2;
        x2 = __umul64hi(x2, a) + x3;
        x3 = __umul64hi(x3, b) + x0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

__global__ void k_mullo64(u64 *out, u64 a0, u64 b0)
{
    u64 a = a0 + threadIdx.x, b = b0 + blockIdx.x;
    u64 x0 = a, x1 = b, x2 = a ^ b, x3 = a + b;
    for (int i = 0; i < ITERS; i++)
    {
        x0 = x0 * a + x1;
        x1 = x1 * b + x2;
        x2 = x2 * a + x3;
        x3 = x3 * b + x0;
    }
    out[blockIdx.x * blockDim.x + threadIdx.x] = x0 ^ x1 ^ x2 ^ x3;
}

// the reference's lazy forward butterfly (ntt.cpp:245-252), 4 independent pairs per lane
__global__ void k_butterfly(u64 *out, u64 w, u64 ws, u64 p)
{
    u64 x[8];
    for (int i = 0; i < 8; i++)
        x[i] = (threadIdx.x * 977u + blockIdx.x * 131u + i) % p;
    const u64 two_p = 2 * p;
    for (int i = 0; i < ITERS / 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            u64 u = x[j], y = x[j + 4];
            u64 q = __umul64hi(y, ws);
            u64 v = y * w - q * p;
            x[j] = u + v;
            x[j + 4] = u - v + two_p;
        }
        w += 2; // keep the compiler from hoisting
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++)
        r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}


// ---- hand-scheduled variants (per-lane twiddles, as in rounds 2.. of the NTT kernels) ----
typedef unsigned u32;
__device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c)
{
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
    return d;
}
__device__ __forceinline__ u64 mul64(u32 a, u32 b)
{
    u64 d, cy;
    asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
    return d;
}
// floor(x*s / 2^64): 4 multiplier ops + cndmask + one move (carry of the middle sum kept instead of splitting it)
__device__ __forceinline__ u64 mulhi_c(u64 x, u64 s)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), s0 = (u32)s, s1 = (u32)(s >> 32);
    const u32 h = __umulhi(x0, s0);
    const u64 A = mad64(x1, s0, (u64)h);
    u64 B;
    u32 cb;
    asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, vcc"
        : "=&v"(B), "=&v"(cb)
        : "v"(x0), "v"(s1), "v"(A)
        : "vcc");
    const u64 addend = (u64)(u32)(B >> 32) | ((u64)cb << 32);
    return mad64(x1, s1, addend);
}
// lo64(x*w + q*np): 6 multiplier ops + 1 add
__device__ __forceinline__ u64 mullo2(u64 x, u64 w, u64 q, u64 np)
{
    const u32 x0 = (u32)x, x1 = (u32)(x >> 32), w0 = (u32)w, w1 = (u32)(w >> 32);
    const u32 q0 = (u32)q, q1 = (u32)(q >> 32), n0 = (u32)np, n1 = (u32)(np >> 32);
    u64 E = mul64(x0, w1);
    E = mad64(x1, w0, E);
    E = mad64(q0, n1, E);
    E = mad64(q1, n0, E);
    u64 V = mul64(x0, w0);
    V = mad64(q0, n0, V);
    u32 vh;
    asm("v_add_u32 %0, %1, %2" : "=v"(vh) : "v"((u32)(V >> 32)), "v"((u32)E));
    return (u64)(u32)V | ((u64)vh << 32);
}

template <int MODE>
__global__ void k_butterfly_lane(u64 *out, u64 w_, u64 ws_, u64 p)
{
    u64 x[8];
    for (int i = 0; i < 8; i++)
        x[i] = (threadIdx.x * 977u + blockIdx.x * 131u + i) % p;
    const u64 two_p = 2 * p, np = 0 - p;
    u64 w = w_ + threadIdx.x, ws = ws_ + 3 * threadIdx.x; // per-lane values (arithmetic content irrelevant here)
    for (int i = 0; i < ITERS / 4; i++)
    {
#pragma unroll
        for (int j = 0; j < 4; j++)
        {
            u64 u = x[j], y = x[j + 4];
            u64 v;
            if (MODE == 0)
            {
                u64 q = __umul64hi(y, ws);
                v = y * w + q * np;
            }
            else
            {
                u64 q = mulhi_c(y, ws);
                v = mullo2(y, w, q, np);
            }
            x[j] = u + v;
            x[j + 4] = u - v + two_p;
        }
        w += 2;
    }
    u64 r = 0;
    for (int i = 0; i < 8; i++)
        r ^= x[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = r;
}

template <class F>
double time_ms(F launch)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    launch();
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < 5; i++)
        launch();
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    return ms / 5;
}

int main()
{
    const int blocks = 256 * 8, threads = 256; // 8 waves/SIMD worth of work per CU
    u64 *out;
    CK(hipMalloc(&out, sizeof(u64) * blocks * threads));
    const double lanes = (double)blocks * threads;
    const u64 p = 36028797017456641ull;
    double ms;
    ms = time_ms([&] { k_mad64<<<blocks, threads>>>(out, 12345u, 6789u); });
    printf("v_mad_u64_u32      : %8.2f Gop/s per lane-op  (%.3f ms)\n", lanes * ITERS * 4 / ms / 1e6, ms);
    ms = time_ms([&] { k_mullo32<<<blocks, threads>>>(out, 12345u, 6789u); });
    printf("v_mul_lo_u32 + add : %8.2f Gop/s              (%.3f ms)\n", lanes * ITERS * 4 / ms / 1e6, ms);
    ms = time_ms([&] { k_mulhi64<<<blocks, threads>>>(out, 0x123456789abcdefull, 0xfedcba987654321ull); });
    printf("mulhi64 (+add)     : %8.2f Gop/s              (%.3f ms)\n", lanes * ITERS * 4 / ms / 1e6, ms);
    ms = time_ms([&] { k_mullo64<<<blocks, threads>>>(out, 0x123456789abcdefull, 0xfedcba987654321ull); });
    printf("mullo64 (+add)     : %8.2f Gop/s              (%.3f ms)\n", lanes * ITERS * 4 / ms / 1e6, ms);
    const u64 w = 1155186985540ull;
    const u64 ws = (u64)((((unsigned __int128)w) << 64) / p);
    ms = time_ms([&] { k_butterfly<<<blocks, threads>>>(out, w, ws, p); });
    const double bf = lanes * ITERS;
    printf("lazy butterfly     : %8.2f G butterflies/s    (%.3f ms)\n", bf / ms / 1e6, ms);
    {
        u64 *o2;
        CK(hipMalloc(&o2, sizeof(u64) * blocks * threads));
        k_butterfly_lane<0><<<blocks, threads>>>(out, w, ws, p);
        k_butterfly_lane<1><<<blocks, threads>>>(o2, w, ws, p);
        CK(hipDeviceSynchronize());
        u64 *h1 = (u64 *)malloc(sizeof(u64) * blocks * threads), *h2 = (u64 *)malloc(sizeof(u64) * blocks * threads);
        CK(hipMemcpy(h1, out, sizeof(u64) * blocks * threads, hipMemcpyDeviceToHost));
        CK(hipMemcpy(h2, o2, sizeof(u64) * blocks * threads, hipMemcpyDeviceToHost));
        size_t bad = 0;
        for (size_t i = 0; i < (size_t)blocks * threads; i++)
            bad += h1[i] != h2[i];
        printf("hand-scheduled butterfly == compiler butterfly on %d lanes: %s (%zu mismatches)\n", blocks * threads, bad ? "NO" : "yes", bad);
        double m0 = time_ms([&] { k_butterfly_lane<0><<<blocks, threads>>>(out, w, ws, p); });
        double m1 = time_ms([&] { k_butterfly_lane<1><<<blocks, threads>>>(out, w, ws, p); });
        printf("per-lane twiddles, compiler   : %8.2f G butterflies/s (%.3f ms)\n", bf / m0 / 1e6, m0);
        printf("per-lane twiddles, hand mads  : %8.2f G butterflies/s (%.3f ms)\n", bf / m1 / 1e6, m1);
        // occupancy sweep: one block of T threads per CU
        for (int t = 256; t <= 1024; t *= 2)
        {
            double a0 = time_ms([&] { k_butterfly_lane<0><<<256, t>>>(out, w, ws, p); });
            double a1 = time_ms([&] { k_butterfly_lane<1><<<256, t>>>(out, w, ws, p); });
            printf("  %d wave(s)/SIMD: compiler %8.2f, hand %8.2f G butterflies/s\n", t / 256, 256.0 * t * ITERS / a0 / 1e6,
                   256.0 * t * ITERS / a1 / 1e6);
        }
    }
    printf("  -> N=2^15 row = 245760 butterflies: ALU bound %.2f M NTT/s = %.1f%% of the 15.26 M/s HBM roofline\n",
           bf / ms / 1e6 * 1e9 / 245760 / 1e6, bf / ms / 1e6 * 1e9 / 245760 / 15.26e6 * 100);
    return 0;
}
