// ubench_issue_occupancy.hip -- round 4: how many waves per SIMD the forward butterfly sequence needs to fill the vector
// issue port. The single-pass transforms hold four waves per SIMD (two workgroups of 512 lanes per CU); while one workgroup
// sits in a load, an exchange or at a barrier only the other's two waves per SIMD have arithmetic to issue. If two waves
// of the lock-step sequence (four independent butterflies interleaved) already run at the four-wave rate, the quarter of
// the issue ceiling the kernels miss is not an ILP problem of the sequence; if they do not, it is.
//
// Same sequences as csrc/ntt.hip's rounds (devmath.hpp), 16 values + four twiddles per lane in registers, no memory.
// Workgroups of 256 lanes (one wave per SIMD each); the dynamic LDS request sets how many are resident per CU.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igemini-seal_amd/csrc -o tools/bin/ubench_issue_occupancy tools/ubench_issue_occupancy.hip
// run  : tools/bin/ubench_issue_occupancy
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

#include "devmath.hpp"

using namespace sealhip;


#define CHECK(x)                                                                                                               \
    do                                                                                                                         \
    {                                                                                                                          \
        hipError_t e_ = (x);                                                                                                   \
        if (e_ != hipSuccess)                                                                                                  \
        {                                                                                                                      \
            std::fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_));                                     \
            std::exit(1);                                                                                                      \
        }                                                                                                                      \
    } while (0)

constexpr int IL = 4;

template <int KIND>
__global__ __launch_bounds__(256) void rate_kernel(u64 *__restrict__ sink, u64 p, int iters)
{
    extern __shared__ u64 lds_unused[];
    u64 x[16], w[IL], ws[IL];
    const u64 seed = (static_cast<u64>(blockIdx.x) * 256 + threadIdx.x) * 0x9E3779B97F4A7C15ull;
#pragma unroll
    for (int i = 0; i < 16; i++)
        x[i] = (seed + static_cast<u64>(i) * 0xBF58476D1CE4E5B9ull) % p;
#pragma unroll
    for (int j = 0; j < IL; j++)
    {
        const u64 wv = (seed ^ (0x94D049BB133111EBull * (j + 1))) % p;
        w[j] = wv;
        ws[j] = static_cast<u64>((static_cast<unsigned __int128>(wv) << 64) / p);
    }
    const u64 neg_p = 0 - p, two_p = 2 * p;
    ZeroHi<2> zp;
    zp.init();
    u64 four_p = two_p << 1;
    asm("" : "+s"(four_p));
    for (int it = 0; it < iters; it++)
    {
#pragma unroll
        for (int W = 3; W >= 0; W--)
        {
#pragma unroll
            for (int c = 0; c < 8; c += IL)
            {
                u64 u[IL], y[IL];
                const int bit = 1 << W;
#pragma unroll
                for (int j = 0; j < IL; j++)
                {
                    const int sl = (((c + j) >> W) << (W + 1)) | ((c + j) & (bit - 1));
                    u[j] = x[sl];
                    y[j] = x[sl | bit];
                }
                if constexpr (KIND == 0)
                    butterflies_fwd_hs<false, IL, 0>(u, y, w, ws, neg_p, two_p);
                else
                    butterflies_fwd_apx2<false, IL>(u, y, w, ws, neg_p, four_p, zp.z);
#pragma unroll
                for (int j = 0; j < IL; j++)
                {
                    const int sl = (((c + j) >> W) << (W + 1)) | ((c + j) & (bit - 1));
                    x[sl] = u[j];
                    x[sl | bit] = y[j];
                }
            }
        }
    }
    u64 acc = 0;
#pragma unroll
    for (int i = 0; i < 16; i++)
        acc ^= x[i];
    if (acc == 0x0123456789ABCDEFull)
        sink[0] = acc + lds_unused[0];
}

template <int KIND>
static double run(int wgs_per_cu, u64 *sink, u64 p)
{
    // 160 KB of LDS per CU: a request of floor(160 / n) KB (less a little) admits exactly n workgroups
    const int lds = (160 * 1024) / wgs_per_cu - 1024;
    CHECK(hipFuncSetAttribute(reinterpret_cast<const void *>(&rate_kernel<KIND>), hipFuncAttributeMaxDynamicSharedMemorySize, lds));
    const unsigned blocks = 256u * wgs_per_cu * 4u;
    const int iters = 800;
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    rate_kernel<KIND><<<blocks, 256, lds>>>(sink, p, iters / 8);
    CHECK(hipEventRecord(a));
    rate_kernel<KIND><<<blocks, 256, lds>>>(sink, p, iters);
    rate_kernel<KIND><<<blocks, 256, lds>>>(sink, p, iters);
    CHECK(hipEventRecord(b));
    CHECK(hipDeviceSynchronize());
    float ms = 0;
    CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return 2.0 * blocks * 256.0 * iters * 32.0 / (ms / 1e3);
}

int main()
{
    const u64 p = 36028797017456641ull; // a 55-bit prime = 1 mod 2^16
    u64 *sink = nullptr;
    CHECK(hipMalloc(&sink, 8));
    std::printf("waves/SIMD  exact-quotient T bf/s  level-2 T bf/s   (butterflies per second, whole device)\n");
    for (int rep = 0; rep < 2; rep++)
        for (int n = 1; n <= 4; n++)
        {
            const double r0 = run<0>(n, sink, p), r2 = run<2>(n, sink, p);
            std::printf("%d           %.3f                 %.3f\n", n, r0 / 1e12, r2 / 1e12);
        }
    CHECK(hipFree(sink));
    return 0;
}
