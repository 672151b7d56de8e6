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
    printf("  -> N=2^15 row = 245760 butterflies: ALU bound %.2f M NTT/s = %.1f%% of the 15.26 M/s HBM roofline\n",
           bf / ms / 1e6 * 1e9 / 245760 / 1e6, bf / ms / 1e6 * 1e9 / 245760 / 15.26e6 * 100);
    return 0;
}
