// ubench_store_pattern.hip -- which final-round store pattern does gfx950 write fastest?
// The single-pass forward NTT at N = 2^15 finishes with each lane holding runs of 4 consecutive coefficients
// (32 bytes); every store instruction writes 16-byte pieces at a 32-byte stride. This benchmark writes the same
// half rows (512 lanes x 32 words per workgroup, two workgroups per row) with nothing but the stores, in the
// lane -> address patterns the candidate in-register transpositions would produce, plain and nontemporal.
// Build: hipcc --offload-arch=gfx950 -O3 tools/ubench_store_pattern.hip -o tools/bin/ubench_store_pattern
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>

typedef unsigned long long u64;
typedef u64 u64x2 __attribute__((ext_vector_type(2)));
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

// pattern ids
//  0  contiguous: 16 B per lane, lanes consecutive (what N = 2^14 has; a whole workgroup writes 8 KiB per instruction)
//  1  f2: 16 B at a 32-byte stride, second instruction fills the holes (today's N = 2^15 pattern)
//  2  pair: lanes (2i, 2i+1) write 32 contiguous bytes, 64-byte stride between pairs
//  3  quad: lanes (4i..4i+3) write 64 contiguous bytes, 128-byte stride between quads (DPP quad_perm transposition)
//  4  half-wave interleave: lane r and lane r+32 write adjacent 16-byte pieces (permlane32_swap transposition)
//  5  wave-contiguous: each wave writes one contiguous KiB per instruction (full transposition)
//  6  row16: lanes (16i..16i+15) write 256 contiguous bytes, 512-byte stride (DPP row transposition)
template <int PAT, bool NT>
__global__ __launch_bounds__(512) void k_store(u64 *__restrict__ out, u64 seed)
{
    const int tid = threadIdx.x;
    u64 *half = out + (size_t)blockIdx.x * 16384;
    u64 v = seed + tid;
#pragma unroll
    for (int F = 0; F < 8; F++)
#pragma unroll
        for (int e = 0; e < 2; e++)
        {
            int c;
            const int w = tid >> 6, l = tid & 63;
            if (PAT == 0)
                c = (F * 2 + e) * 1024 + tid * 2;
            else if (PAT == 1)
                c = F * 2048 + tid * 4 + e * 2;
            else if (PAT == 2)
                c = F * 2048 + (tid >> 1) * 8 + (tid & 1) * 2 + e * 4;
            else if (PAT == 3)
                c = F * 2048 + (tid >> 2) * 16 + (tid & 3) * 2 + e * 8;
            else if (PAT == 4)
                c = F * 2048 + w * 256 + (l & 31) * 4 + (l >> 5) * 2 + e * 128;
            else if (PAT == 5)
                c = F * 2048 + w * 256 + l * 2 + e * 128;
            else
                c = F * 2048 + (tid >> 4) * 64 + (tid & 15) * 2 + e * 32;
            u64x2 val;
            val.x = v + F;
            val.y = v ^ e;
            if (NT)
                __builtin_nontemporal_store(val, reinterpret_cast<u64x2 *>(half + c));
            else
                *reinterpret_cast<u64x2 *>(half + c) = val;
        }
}

template <int PAT, bool NT>
static void run(u64 *buf, int rows, const char *name)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    const int reps = 20;
    for (int i = 0; i < 3; i++)
        k_store<PAT, NT><<<rows * 2, 512>>>(buf, i);
    CK(hipDeviceSynchronize());
    CK(hipEventRecord(a));
    for (int i = 0; i < reps; i++)
        k_store<PAT, NT><<<rows * 2, 512>>>(buf, i);
    CK(hipEventRecord(b));
    CK(hipEventSynchronize(b));
    float ms;
    CK(hipEventElapsedTime(&ms, a, b));
    const double bytes = (double)rows * 32768 * 8 * reps;
    printf("%-28s %-4s %7.3f ms/launch  %6.2f TB/s\n", name, NT ? "nt" : "", ms / reps, bytes / (ms * 1e-3) / 1e12);
}

int main(int argc, char **argv)
{
    const int rows = argc > 1 ? atoi(argv[1]) : 8064;
    u64 *buf;
    CK(hipMalloc(&buf, (size_t)rows * 32768 * 8));
    printf("store-only, %d rows of 2^15 words (%.1f MB), 2 workgroups of 512 lanes per row\n", rows, rows * 32768.0 * 8 / 1e6);
    run<0, false>(buf, rows, "0 contiguous");
    run<0, true>(buf, rows, "0 contiguous");
    run<1, false>(buf, rows, "1 f2 16B@32B");
    run<1, true>(buf, rows, "1 f2 16B@32B");
    run<2, false>(buf, rows, "2 pair 32B@64B");
    run<2, true>(buf, rows, "2 pair 32B@64B");
    run<3, false>(buf, rows, "3 quad 64B@128B");
    run<3, true>(buf, rows, "3 quad 64B@128B");
    run<4, false>(buf, rows, "4 halfwave interleave");
    run<4, true>(buf, rows, "4 halfwave interleave");
    run<5, false>(buf, rows, "5 wave-contiguous 1KiB");
    run<5, true>(buf, rows, "5 wave-contiguous 1KiB");
    run<6, false>(buf, rows, "6 row16 256B@512B");
    run<6, true>(buf, rows, "6 row16 256B@512B");
    CK(hipFree(buf));
    return 0;
}
