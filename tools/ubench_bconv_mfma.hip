// ubench_bconv_mfma.hip -- round 4, VERDICT r03 item 5: the byte-limb int8-MFMA form of a BEHZ base conversion TIMED
// against the carry-free vector-ALU form the engine ships (devmath.hpp DotAcc), instead of estimated.
//
// One instance, the one the verdict names: config 3's q -> Bsk conversion inside bfv_lift2 / bfv_floor_sk2
// (BaseConverter::fast_convert_array, native/src/seal/util/rns.cpp:469-496): per coefficient column 7 words t_i < 2^56 in,
// 8 words out,  r_j = (sum_i t_i * M_ji) mod p_j  with M a constant 8 x 7 matrix of residues of 60-bit primes p_j.
// Both kernels compute that canonical residue from the exact integer sum (so they agree bit for bit, and with the
// reference's sum-of-exact-products-then-one-reduction), reduce it with the same Montgomery step, read the same 7 rows and
// write the same 8 rows; only how the 56 products per column are formed differs.
//
//   valu : one lane per column, DotAcc<7> per output (five v_mad_u64_u32 per product, no carries), as in csrc/rns.hip.
//   mfma : one wave per 64 columns. The K dimension of v_mfma_i32_16x16x64_i8 is (word i, byte a) = 7 x 8 = 56 <= 64, so the
//          B operand of a 16-column tile is the raw little-endian bytes of the words: lane (column c, k-block kb) loads
//          words 2 kb and 2 kb + 1 of its column -- no byte splitting at all. The bytes are read as SIGNED int8, so every
//          data dword is xor-ed with 0x80808080 (u = s + 128) and the accumulator starts at 128 * (sum of the constant
//          digits of its row), which is a constant of the context. The A operand of output j is the 16 x 64 Toeplitz
//          matrix A[s][(i, a)] = d_ji[s - a] of the balanced base-256 digits d in [-128, 127] of M_ji (host-side table), so
//          D[s][c] = S_{j,s} = sum over a + b = s of (byte a of t_i) * d_ji[b]: the limb sums, |S| < 2^20. A lane ends up
//          with four consecutive limb sums of one (output, column) per MFMA; it folds them into one signed 64-bit partial
//          (three multiply-adds), a 4 x 4 transposition across the four 16-lane rows (v_permlane32_swap / v_permlane16_swap,
//          eight moves per four outputs) brings the four partials of an output into one lane, which assembles the signed
//          128-bit sum and reduces it. 32 MFMAs per 64 columns: ~1 % of the int8 rate, as the verdict says; what is timed
//          here is everything else.
//
// build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -Igemini-seal_amd/csrc -o tools/bin/ubench_bconv_mfma tools/ubench_bconv_mfma.hip
// run  : tools/bin/ubench_bconv_mfma [items of 32768 columns, default 512]
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "devmath.hpp"

using namespace sealhip;
typedef unsigned __int128 u128;
typedef __int128 i128;
typedef int v4i __attribute__((ext_vector_type(4)));

constexpr int KIN = 7, NOUT = 8;
struct Consts
{
    u64 p[NOUT], ninv[NOUT], rdp[NOUT];
    u64 m[NOUT][KIN]; // M_ji * 2^64 mod p_j (the Montgomery step takes the factor out again)
};
__constant__ Consts g_c;

#define CK(x)                                                                             \
    do                                                                                    \
    {                                                                                     \
        hipError_t e_ = (x);                                                              \
        if (e_ != hipSuccess)                                                             \
        {                                                                                 \
            std::printf("HIP error %s at %s:%d\n", hipGetErrorString(e_), __FILE__, __LINE__); \
            std::exit(1);                                                                 \
        }                                                                                 \
    } while (0)

// ---------------------------------------------------------------- the shipped form
// REPS > 1: the conversion is repeated on the loaded words (each repetition on t + rep, results xor-ed) so that the kernels are
// bound by their arithmetic, as the fused BEHZ kernels are (bfv_floor_sk2 forms ~170 products per column from 30 loaded words;
// one conversion alone -- 56 products from 7 words -- is bound by its memory traffic in either form)
template <int REPS>
__global__ __launch_bounds__(256) void bconv_valu(const u64 *__restrict__ in, u64 *__restrict__ out, std::size_t ncols)
{
    const std::size_t c = blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x;
    if (c >= ncols)
        return;
    u64 t[KIN];
#pragma unroll
    for (int i = 0; i < KIN; i++)
        t[i] = in[i * ncols + c];
    u64 res[NOUT] = {};
#pragma unroll
    for (int rep = 0; rep < REPS; rep++)
    {
    SplitT ts[KIN];
#pragma unroll
    for (int i = 0; i < KIN; i++)
        ts[i] = SplitT(t[i] + rep);
#pragma unroll
    for (int j = 0; j < NOUT; j++)
    {
        DotAcc<KIN> acc;
#pragma unroll
        for (int i = 0; i < KIN; i++)
        {
            if (i == 0) acc.add<0>(ts[0], g_c.m[j][0]);
            if (i == 1) acc.add<1>(ts[1], g_c.m[j][1]);
            if (i == 2) acc.add<2>(ts[2], g_c.m[j][2]);
            if (i == 3) acc.add<3>(ts[3], g_c.m[j][3]);
            if (i == 4) acc.add<4>(ts[4], g_c.m[j][4]);
            if (i == 5) acc.add<5>(ts[5], g_c.m[j][5]);
            if (i == 6) acc.add<6>(ts[6], g_c.m[j][6]);
        }
        u64 lo, hi;
        acc.finish(lo, hi);
        res[j] ^= redc_finish(redc128(lo, hi, g_c.p[j], g_c.ninv[j]), g_c.p[j], g_c.rdp[j], false);
    }
    }
#pragma unroll
    for (int j = 0; j < NOUT; j++)
        out[j * ncols + c] = res[j];
}

// ---------------------------------------------------------------- the int8-MFMA form
__device__ __forceinline__ void swap32(long long &a, long long &b) // lanes 32-63 of a <-> lanes 0-31 of b
{
    const u64 ua = static_cast<u64>(a), ub = static_cast<u64>(b);
    const auto lo = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(ua), static_cast<unsigned>(ub), false, false);
    const auto hi = __builtin_amdgcn_permlane32_swap(static_cast<unsigned>(ua >> 32), static_cast<unsigned>(ub >> 32), false, false);
    a = static_cast<long long>(lo[0] | (static_cast<u64>(hi[0]) << 32));
    b = static_cast<long long>(lo[1] | (static_cast<u64>(hi[1]) << 32));
}
__device__ __forceinline__ void swap16(long long &a, long long &b) // odd 16-lane rows of a <-> even rows of b
{
    const u64 ua = static_cast<u64>(a), ub = static_cast<u64>(b);
    const auto lo = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(ua), static_cast<unsigned>(ub), false, false);
    const auto hi = __builtin_amdgcn_permlane16_swap(static_cast<unsigned>(ua >> 32), static_cast<unsigned>(ub >> 32), false, false);
    a = static_cast<long long>(lo[0] | (static_cast<u64>(hi[0]) << 32));
    b = static_cast<long long>(lo[1] | (static_cast<u64>(hi[1]) << 32));
}

// atab: [NOUT][64 lanes] v4i (16 signed digits), cinit: [NOUT][64 lanes] v4i (accumulator start values)
template <int REPS>
__global__ __launch_bounds__(256) void bconv_mfma(const u64 *__restrict__ in, u64 *__restrict__ out, std::size_t ncols,
                                                  const v4i *__restrict__ atab, const v4i *__restrict__ cinit)
{
    const int lane = threadIdx.x & 63;
    const std::size_t wave = (blockIdx.x * static_cast<std::size_t>(blockDim.x) + threadIdx.x) >> 6;
    const std::size_t nwaves = (static_cast<std::size_t>(gridDim.x) * blockDim.x) >> 6;
    const int cl = lane & 15, g = lane >> 4;
    v4i A[NOUT], C0[NOUT];
#pragma unroll
    for (int j = 0; j < NOUT; j++)
    {
        A[j] = atab[j * 64 + lane];
        C0[j] = cinit[j * 64 + lane];
    }
    const u64 pj0 = g_c.p[g], pj1 = g_c.p[g + 4]; // this lane finishes outputs g and g + 4
    const u64 nv0 = g_c.ninv[g], nv1 = g_c.ninv[g + 4], rd0 = g_c.rdp[g], rd1 = g_c.rdp[g + 4];
    for (std::size_t base = wave * 64; base < ncols; base += nwaves * 64)
    {
#pragma unroll
        for (int nt = 0; nt < 4; nt++)
        {
            const std::size_t col = base + nt * 16 + cl;
            // B operand: the raw bytes of words 2g and 2g + 1 of this column (word 7 does not exist: zero)
            const u64 w0l = in[(2 * g) * ncols + col];
            const u64 w1l = g < 3 ? in[(2 * g + 1) * ncols + col] : 0;
            u64 res0 = 0, res1 = 0;
#pragma unroll
            for (int rep = 0; rep < REPS; rep++)
            {
            const u64 w0 = w0l + rep, w1 = g < 3 ? w1l + rep : 0;
            v4i B;
            B[0] = static_cast<int>(static_cast<unsigned>(w0) ^ 0x80808080u);
            B[1] = static_cast<int>(static_cast<unsigned>(w0 >> 32) ^ 0x80808080u);
            B[2] = static_cast<int>(static_cast<unsigned>(w1) ^ 0x80808080u);
            B[3] = static_cast<int>(static_cast<unsigned>(w1 >> 32) ^ 0x80808080u);
            long long P[NOUT];
#pragma unroll
            for (int j = 0; j < NOUT; j++)
            {
                const v4i S = __builtin_amdgcn_mfma_i32_16x16x64_i8(A[j], B, C0[j], 0, 0, 0);
                // limb sums 4g .. 4g + 3 of (output j, this column) -> one signed partial, weight 2^(32 g)
                P[j] = static_cast<long long>(S[0]) + (static_cast<long long>(S[1]) << 8) + (static_cast<long long>(S[2]) << 16) +
                       (static_cast<long long>(S[3]) << 24);
            }
            // 4 x 4 transposition across the four 16-lane rows, twice (outputs 0-3 and 4-7): afterwards P[4h + r] of lane
            // row g is the partial of weight 2^(32 r) of output 4h + g
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
                swap32(P[4 * h + 0], P[4 * h + 2]);
                swap32(P[4 * h + 1], P[4 * h + 3]);
                swap16(P[4 * h + 0], P[4 * h + 1]);
                swap16(P[4 * h + 2], P[4 * h + 3]);
            }
#pragma unroll
            for (int h = 0; h < 2; h++)
            {
                const i128 v = static_cast<i128>(P[4 * h]) + (static_cast<i128>(P[4 * h + 1]) << 32) +
                               (static_cast<i128>(P[4 * h + 2]) << 64) + (static_cast<i128>(P[4 * h + 3]) << 96);
                const u64 lo = static_cast<u64>(static_cast<u128>(v)), hi = static_cast<u64>(static_cast<u128>(v) >> 64);
                const u64 p = h ? pj1 : pj0, nv = h ? nv1 : nv0, rd = h ? rd1 : rd0;
                (h ? res1 : res0) ^= redc_finish(redc128(lo, hi, p, nv), p, rd, false);
            }
            }
            out[g * ncols + col] = res0;
            out[(4 + g) * ncols + col] = res1;
        }
    }
}

static u64 mulmod(u64 a, u64 b, u64 p)
{
    return static_cast<u64>(static_cast<u128>(a) * b % p);
}

int main(int argc, char **argv)
{
    const std::size_t items = argc > 1 ? std::strtoull(argv[1], nullptr, 10) : 512;
    const std::size_t ncols = items * 32768;
    // get_primes(32768, 60, .)-like moduli (the auxiliary primes of config 3) and random matrix entries
    const u64 primes[NOUT] = { 1152921504597016577ull, 1152921504595968001ull, 1152921504595640321ull, 1152921504593412097ull,
                               1152921504592822273ull, 1152921504592429057ull, 1152921504589938689ull, 1152921504606584833ull };
    std::mt19937_64 rng(4);
    Consts hc{};
    std::vector<u64> M(NOUT * KIN);
    for (int j = 0; j < NOUT; j++)
    {
        const u64 p = primes[j];
        hc.p[j] = p;
        u64 inv = 1; // -p^-1 mod 2^64 by Newton
        for (int it = 0; it < 6; it++)
            inv *= 2 - p * inv;
        hc.ninv[j] = 0 - inv;
        hc.rdp[j] = static_cast<u64>((static_cast<u128>(1) << 64) / p);
        const u64 r64 = static_cast<u64>((static_cast<u128>(1) << 64) % p);
        for (int i = 0; i < KIN; i++)
        {
            M[j * KIN + i] = rng() % p;
            hc.m[j][i] = mulmod(M[j * KIN + i], r64, p);
        }
    }
    CK(hipMemcpyToSymbol(HIP_SYMBOL(g_c), &hc, sizeof(hc)));
    // balanced base-256 digits of the (Montgomery-scaled) constants, the Toeplitz A operands and the accumulator start values
    std::vector<int> atab(NOUT * 64 * 4), cinit(NOUT * 64 * 4);
    for (int j = 0; j < NOUT; j++)
    {
        int d[KIN][9];
        for (int i = 0; i < KIN; i++)
        {
            u64 v = hc.m[j][i];
            int carry = 0;
            for (int b = 0; b < 9; b++)
            {
                int x = static_cast<int>(v & 0xFF) + carry;
                v >>= 8;
                carry = 0;
                if (x >= 128)
                {
                    x -= 256;
                    carry = 1;
                }
                d[i][b] = x;
            }
            if (d[i][8] != 0)
            {
                std::printf("constant does not fit eight balanced digits\n");
                return 1;
            }
        }
        for (int lane = 0; lane < 64; lane++)
        {
            const int s = lane & 15, kb = lane >> 4; // A: row s, k-block kb
            unsigned char bytes[16];
            for (int e = 0; e < 16; e++)
            {
                const int i = 2 * kb + e / 8, a = e % 8, b = s - a;
                bytes[e] = static_cast<unsigned char>((i < KIN && b >= 0 && b <= 7) ? d[i][b] : 0);
            }
            for (int r = 0; r < 4; r++)
                atab[(j * 64 + lane) * 4 + r] = static_cast<int>(bytes[4 * r] | (bytes[4 * r + 1] << 8) | (bytes[4 * r + 2] << 16) |
                                                                 (static_cast<unsigned>(bytes[4 * r + 3]) << 24));
            for (int r = 0; r < 4; r++) // C/D: row 4 (lane >> 4) + r, column lane & 15
            {
                const int row = 4 * (lane >> 4) + r;
                long long sum = 0;
                for (int i = 0; i < KIN; i++)
                    for (int a = 0; a < 8; a++)
                        if (row - a >= 0 && row - a <= 7)
                            sum += d[i][row - a];
                cinit[(j * 64 + lane) * 4 + r] = static_cast<int>(128 * sum);
            }
        }
    }
    std::vector<u64> hin(static_cast<std::size_t>(KIN) * 65536);
    u64 *din, *dout_a, *dout_b;
    int *datab, *dcinit;
    CK(hipMalloc(&din, KIN * ncols * 8));
    CK(hipMalloc(&dout_a, NOUT * ncols * 8));
    CK(hipMalloc(&dout_b, NOUT * ncols * 8));
    CK(hipMalloc(&datab, atab.size() * 4));
    CK(hipMalloc(&dcinit, cinit.size() * 4));
    CK(hipMemcpy(datab, atab.data(), atab.size() * 4, hipMemcpyHostToDevice));
    CK(hipMemcpy(dcinit, cinit.data(), cinit.size() * 4, hipMemcpyHostToDevice));
    // inputs: residues below 2^56 (the 55-bit q primes of config 3), the first columns adversarial (all-ones bytes, zeros)
    std::vector<u64> blockv(1 << 20);
    for (int i = 0; i < KIN; i++)
        for (std::size_t off = 0; off < ncols; off += blockv.size())
        {
            for (std::size_t c = 0; c < blockv.size(); c++)
                blockv[c] = rng() >> 8;
            if (off == 0)
            {
                blockv[0] = (u64(1) << 56) - 1;
                blockv[1] = 0;
                blockv[2] = 0x0080808080808080ull;
                blockv[3] = 0x007F7F7F7F7F7F7Full;
            }
            CK(hipMemcpy(din + i * ncols + off, blockv.data(), std::min(blockv.size(), ncols - off) * 8, hipMemcpyHostToDevice));
        }
    const unsigned vblocks = static_cast<unsigned>((ncols + 255) / 256);
    const unsigned mblocks = 256 * 8; // persistent waves: 8192 waves walk the columns in groups of 64
    hipEvent_t e0, e1;
    CK(hipEventCreate(&e0));
    CK(hipEventCreate(&e1));
    auto time_it = [&](auto launch) {
        launch();
        CK(hipDeviceSynchronize());
        CK(hipEventRecord(e0));
        for (int r = 0; r < 10; r++)
            launch();
        CK(hipEventRecord(e1));
        CK(hipDeviceSynchronize());
        float ms = 0;
        CK(hipEventElapsedTime(&ms, e0, e1));
        return ms / 10;
    };
    const float t_valu4 = time_it([&] { bconv_valu<4><<<vblocks, 256>>>(din, dout_a, ncols); });
    const float t_mfma4 = time_it([&] {
        bconv_mfma<4><<<mblocks, 256>>>(din, dout_b, ncols, reinterpret_cast<const v4i *>(datab), reinterpret_cast<const v4i *>(dcinit));
    });
    const float t_valu = time_it([&] { bconv_valu<1><<<vblocks, 256>>>(din, dout_a, ncols); });
    const float t_mfma = time_it([&] {
        bconv_mfma<1><<<mblocks, 256>>>(din, dout_b, ncols, reinterpret_cast<const v4i *>(datab), reinterpret_cast<const v4i *>(dcinit));
    });
    CK(hipGetLastError());
    // bit-exactness: the two kernels against each other on every word, and against __int128 on the first columns
    std::vector<u64> ha(NOUT * 4096), hb(NOUT * 4096), hin0(KIN * 4096);
    std::size_t mismatches = 0;
    for (int j = 0; j < NOUT; j++)
        for (std::size_t off = 0; off < ncols; off += std::size_t(1) << 22)
        {
            const std::size_t nn = std::min<std::size_t>(std::size_t(1) << 22, ncols - off);
            std::vector<u64> xa(nn), xb(nn);
            CK(hipMemcpy(xa.data(), dout_a + j * ncols + off, nn * 8, hipMemcpyDeviceToHost));
            CK(hipMemcpy(xb.data(), dout_b + j * ncols + off, nn * 8, hipMemcpyDeviceToHost));
            for (std::size_t c = 0; c < nn; c++)
                mismatches += xa[c] != xb[c];
            if (off == 0)
                for (int c = 0; c < 4096; c++)
                    ha[j * 4096 + c] = xa[c];
        }
    for (int i = 0; i < KIN; i++)
        CK(hipMemcpy(hin0.data() + i * 4096, din + i * ncols, 4096 * 8, hipMemcpyDeviceToHost));
    std::size_t wrong = 0;
    for (int c = 0; c < 4096; c++)
        for (int j = 0; j < NOUT; j++)
        {
            u128 acc = 0; // the reference's order of operations: exact products summed, one reduction
            for (int i = 0; i < KIN; i++)
                acc += static_cast<u128>(hin0[i * 4096 + c]) * M[j * KIN + i];
            wrong += static_cast<u64>(acc % primes[j]) != ha[j * 4096 + c];
        }
    const double bytes = static_cast<double>(KIN + NOUT) * ncols * 8;
    std::printf("q->Bsk conversion, 7 words in / 8 words out per column, %zu columns (%.2f GB moved per launch)\n", ncols, bytes / 1e9);
    std::printf("valu (DotAcc<7>, shipped form): %.3f ms  = %.2f TB/s, %.2f G columns/s\n", t_valu, bytes / t_valu / 1e9, ncols / t_valu / 1e6);
    std::printf("mfma (int8 byte limbs)        : %.3f ms  = %.2f TB/s, %.2f G columns/s   (%.1f %% of the valu form's time)\n", t_mfma,
                bytes / t_mfma / 1e9, ncols / t_mfma / 1e6, 100.0 * t_mfma / t_valu);
    std::printf("the same with the conversion repeated 4 x on the loaded words (arithmetic-bound, like the fused kernels):\n");
    std::printf("valu x4: %.3f ms = %.2f G conversions/s;  mfma x4: %.3f ms = %.2f G conversions/s  (%.1f %% of the valu form's time)\n", t_valu4,
                4.0 * ncols / t_valu4 / 1e6, t_mfma4, 4.0 * ncols / t_mfma4 / 1e6, 100.0 * t_mfma4 / t_valu4);
    std::printf("mfma vs valu: %zu words differ of %zu; valu vs __int128 on 4096 columns: %zu wrong\n", mismatches,
                static_cast<std::size_t>(NOUT) * ncols, wrong);
    return (mismatches || wrong) ? 1 : 0;
}
