// ubench_latency.hip -- issue interval and dependent-issue latency of the integer instructions the NTT butterflies
// use, measured with ONE wave per CU (no other wave to hide anything) and with 2/4/8 waves per SIMD.
// Build: hipcc --offload-arch=gfx950 -O3 -o ubench_latency ubench_latency.hip
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
typedef unsigned u32;
#define CK(x)                                                                                     \
    do                                                                                            \
    {                                                                                             \
        hipError_t e_ = (x);                                                                      \
        if (e_ != hipSuccess)                                                                     \
        {                                                                                         \
            printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__);         \
            exit(1);                                                                              \
        }                                                                                         \
    } while (0)

constexpr int ITERS = 2048;

#define MAD(d, a, b, c) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %3" : "=v"(d) : "v"(a), "v"(b), "v"(c) : "vcc")
#define ADD64(d, a, b) asm volatile("v_lshl_add_u64 %0, %1, 0, %2" : "=v"(d) : "v"(a), "v"(b))
#define MULLO(d, a, b) asm volatile("v_mul_lo_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define MULHI(d, a, b) asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))
#define ADD32(d, a, b) asm volatile("v_add_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b))

// KIND: 0 mad dependent chain (1 chain), 1 mad 2 chains, 2 mad 4 chains, 3 mad 8 chains,
//       4 add64 dependent, 5 add64 4 chains, 6 mul_lo dependent, 7 mul_lo 4 chains, 8 add32 dependent, 9 add32 4 chains
template <int KIND>
__global__ void k(u64 *out, u64 *cycles, u32 a0)
{
    u32 a = a0 + threadIdx.x, b = a0 * 3 + blockIdx.x;
    u64 c[8];
    u32 r[8];
    for (int i = 0; i < 8; i++)
    {
        c[i] = threadIdx.x + i;
        r[i] = threadIdx.x * 7 + i;
    }
    const u64 t0 = __builtin_readcyclecounter();
    for (int it = 0; it < ITERS; it++)
    {
        if (KIND <= 3)
        {
            constexpr int NCH = 1 << KIND;
#pragma unroll
            for (int rep = 0; rep < 8 / NCH; rep++)
#pragma unroll
                for (int j = 0; j < NCH; j++)
                    MAD(c[j], a, b, c[j]);
        }
        else if (KIND == 4 || KIND == 5)
        {
            constexpr int NCH = KIND == 4 ? 1 : 4;
#pragma unroll
            for (int rep = 0; rep < 8 / NCH; rep++)
#pragma unroll
                for (int j = 0; j < NCH; j++)
                    ADD64(c[j], c[j], c[7]);
        }
        else if (KIND == 6 || KIND == 7)
        {
            constexpr int NCH = KIND == 6 ? 1 : 4;
#pragma unroll
            for (int rep = 0; rep < 8 / NCH; rep++)
#pragma unroll
                for (int j = 0; j < NCH; j++)
                    MULLO(r[j], r[j], a);
        }
        else
        {
            constexpr int NCH = KIND == 8 ? 1 : 4;
#pragma unroll
            for (int rep = 0; rep < 8 / NCH; rep++)
#pragma unroll
                for (int j = 0; j < NCH; j++)
                    ADD32(r[j], r[j], a);
        }
    }
    const u64 t1 = __builtin_readcyclecounter();
    u64 x = 0;
    for (int i = 0; i < 8; i++)
        x ^= c[i] ^ r[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = x;
    if (threadIdx.x == 0 && blockIdx.x == 0)
        *cycles = t1 - t0;
}

template <int KIND>
void run(const char *name, u64 *out, u64 *dcyc)
{
    // waves per SIMD: 1 wave per CU .. 8 waves per SIMD
    const int cfg[5][2] = {{256, 64}, {256, 256}, {256, 512}, {256, 1024}, {512, 1024}};
    const char *label[5] = {"1 wave/CU", "1 wave/SIMD", "2 waves/SIMD", "4 waves/SIMD", "8 waves/SIMD"};
    printf("%-28s", name);
    for (int i = 0; i < 5; i++)
    {
        hipEvent_t e0, e1;
        CK(hipEventCreate(&e0));
        CK(hipEventCreate(&e1));
        k<KIND><<<cfg[i][0], cfg[i][1]>>>(out, dcyc, 12345u);
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        k<KIND><<<cfg[i][0], cfg[i][1]>>>(out, dcyc, 12345u);
        CK(hipEventRecord(e1));
        CK(hipEventSynchronize(e1));
        float ms;
        CK(hipEventElapsedTime(&ms, e0, e1));
        u64 cyc;
        CK(hipMemcpy(&cyc, dcyc, 8, hipMemcpyDeviceToHost));
        const double per_instr_cycles = (double)cyc / (ITERS * 8.0);
        const double waves_per_simd = (double)cfg[i][0] * cfg[i][1] / 64 / 1024.0;
        const double ns_per_instr = ms * 1e6 / (ITERS * 8.0);
        (void)label;
        // cycles per instruction seen by one wave; SIMD issue interval = that / waves per SIMD (if >= 1 wave/SIMD)
        printf(" | %5.1f cyc (%4.1f ns)", per_instr_cycles, ns_per_instr);
        (void)waves_per_simd;
    }
    printf("\n");
}

int main()
{
    u64 *out, *dcyc;
    CK(hipMalloc(&out, 8 * 512 * 1024));
    CK(hipMalloc(&dcyc, 8));
    printf("cycles per instruction as seen by one wave (s_memtime) and ns per instruction (events)\n");
    printf("%-28s | 1 wave/CU          | 1 wave/SIMD        | 2 waves/SIMD       | 4 waves/SIMD       | 8 waves/SIMD\n", "");
    run<0>("v_mad_u64_u32 dependent", out, dcyc);
    run<1>("v_mad_u64_u32 2 chains", out, dcyc);
    run<2>("v_mad_u64_u32 4 chains", out, dcyc);
    run<3>("v_mad_u64_u32 8 chains", out, dcyc);
    run<4>("v_lshl_add_u64 dependent", out, dcyc);
    run<5>("v_lshl_add_u64 4 chains", out, dcyc);
    run<6>("v_mul_lo_u32 dependent", out, dcyc);
    run<7>("v_mul_lo_u32 4 chains", out, dcyc);
    run<8>("v_add_u32 dependent", out, dcyc);
    run<9>("v_add_u32 4 chains", out, dcyc);
    return 0;
}
