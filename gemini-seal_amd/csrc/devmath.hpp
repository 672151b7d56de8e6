// devmath.hpp -- 64-bit modular arithmetic for gfx950 device code.
//
// CDNA4 has no 64x64 multiplier: a 64-bit product is built from v_mad_u64_u32 / v_mul_lo_u32.
// The helpers below state the exact functions of the reference's scalar primitives
// (SURVEY Appendix A.1); where the reference returns a canonical residue any exact reduction
// yields identical bits, where it returns a "lazy" representative the operation sequence is kept.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>
#include "ntt_bounds.hpp"

namespace sealhip
{
    using u64 = unsigned long long;

    // Per-prime constants kept in a device array (one entry per key prime and per auxiliary prime).
    struct PrimeDev
    {
        u64 p;
        u64 two_p;
        u64 rdp;        // floor(2^64 / p)           (ntt.cpp:75 reduce_precomp_)
        u64 cr0, cr1;   // floor(2^128 / p)          (modulus.cpp:85-96)
        u64 inv_n, inv_n_shoup;
        u64 inv_n_w, inv_n_w_shoup; // top inverse-layer twiddle times n^{-1} (ntt.cpp:97)
        u64 ninv;                   // -p^{-1} mod 2^64 (Montgomery reduction of 128-bit dot products)
        const u64 *fwd;  // N {w, w'} pairs, bit-reversed exponent order
        const u64 *inv;  // N {w, w'} pairs for psi^{-1}
        // primes below 2^50 only (else null): the same twiddles as doubles, for the floating-point transform
        const double *fwd_d; // N entries w
        const double *inv_d; // N entries
        double p_d, pinv_d;  // p and the correctly rounded 1/p
    };

    // ---- residues as doubles (primes below 2^50). A product y*w with |y| < 2^53 and 0 <= w < p is formed exactly as
    // h + l (h = the rounded product, l = fma(y, w, -h) its rounding error, both exact integers); with u = 2^-53 the
    // quotient estimate q = rint(fl(h * fl(1/p))) is within 1/2 + |h/p| (2u + u^2) of h/p, h - q*p is an integer below 2^53
    // in magnitude (so the fma forms it exactly) and r = (h - q*p) + l satisfies
    //     r == y*w (mod p),   |r| <= (1/2 + 3 * 2^-53 |y|) p        (at most 3.5 p at |y| = 2^53)
    // -- the rounding of h * (1/p) AND the exact error l both count (ntt_bounds.hpp section 4; round 2 had assumed
    // (1/2 + 2^-52 |y|) p, which adversarial operands exceed: tests/bounds_check.cpp). Every operation is an exact integer
    // computation as long as magnitudes stay below 2^53 = 8 * 2^50, which the callers' reduction schedules guarantee by a
    // worst-case recurrence -- the transform is bit-for-bit the integer one after the final canonicalisation.
    // (v_fma_f64 issues at the rate of v_mad_u64_u32; the modular product takes 6 of them against 14.)
    constexpr double kTwo52 = 4503599627370496.0;
    __device__ __forceinline__ double fp_of(u64 bits)
    {
        return __longlong_as_double(static_cast<long long>(bits));
    }
    __device__ __forceinline__ u64 fp_bits(double v)
    {
        return static_cast<u64>(__double_as_longlong(v));
    }
    // integer below 2^52 -> double (two instructions: or the exponent of 2^52 in, subtract 2^52)
    __device__ __forceinline__ double fp_from_u64(u64 x)
    {
        return fp_of(x | 0x4330000000000000ull) - kTwo52;
    }
    // double holding an integer in [0, 2^52) -> integer
    __device__ __forceinline__ u64 fp_to_u64(double r)
    {
        return fp_bits(r + kTwo52) & 0x000FFFFFFFFFFFFFull;
    }
    // |x| < 2^53 -> the representative in [-p/2, p/2]
    __device__ __forceinline__ double fp_reduce(double x, double p, double pinv)
    {
#pragma clang fp contract(off)
        return __builtin_fma(-__builtin_rint(x * pinv), p, x);
    }
    // -> the canonical residue in [0, p)
    __device__ __forceinline__ double fp_canonical(double x, double p, double pinv)
    {
        const double r = fp_reduce(x, p, pinv);
        return r < 0.0 ? r + p : r;
    }
    __device__ __forceinline__ double fp_mulmod(double y, double w, double p, double pinv)
    {
#pragma clang fp contract(off)
        const double h = y * w;
        const double l = __builtin_fma(y, w, -h);
        const double q = __builtin_rint(h * pinv);
        return __builtin_fma(-q, p, h) + l;
    }
    // Cooley-Tukey butterfly (u, y) -> (u + y*w, u - y*w)
    __device__ __forceinline__ void fp_butterfly_fwd(u64 &ub, u64 &yb, u64 wb, double p, double pinv)
    {
        const double u = fp_of(ub), r = fp_mulmod(fp_of(yb), fp_of(wb), p, pinv);
        ub = fp_bits(u + r);
        yb = fp_bits(u - r);
    }
    // Gentleman-Sande butterfly (u, y) -> (u + y, (u - y)*w)
    __device__ __forceinline__ void fp_butterfly_inv(u64 &ub, u64 &yb, u64 wb, double p, double pinv)
    {
        const double u = fp_of(ub), y = fp_of(yb);
        ub = fp_bits(u + y);
        yb = fp_bits(fp_mulmod(u - y, fp_of(wb), p, pinv));
    }

    // Transparency sink (Ciphertext::is_transparent, ciphertext.h:471-476; checked after every Evaluator operation when
    // SEAL_THROW_ON_TRANSPARENT_CIPHERTEXT is defined, evaluator.cpp:265-271): tflags[item] becomes non-zero iff some word of
    // polynomials 1.. of the item's result is non-zero. The LAST kernel of multiply / relinearize / apply_galois notes it
    // for the words it stores anyway -- no separate read pass over the result. `nz` is the OR of the lane's words of
    // polynomials 1..; the flag is only written while it is still zero (benign race: every writer stores 1).
    __device__ __forceinline__ void note_nonzero(unsigned *__restrict__ tflags, std::size_t item, u64 nz)
    {
        if (tflags != nullptr && nz != 0 && tflags[item] == 0)
            tflags[item] = 1;
    }

    // streaming stores (nontemporal hint): outputs that the producing kernel does not read again should not push the
    // constant tables (twiddles, key slices) out of L2
    __device__ __forceinline__ void store_stream(u64 *p, u64 v)
    {
        __builtin_nontemporal_store(v, p);
    }
    __device__ __forceinline__ void store_stream2(u64 *p, u64 a, u64 b)
    {
        typedef u64 u64x2_s __attribute__((ext_vector_type(2)));
        u64x2_s v;
        v.x = a;
        v.y = b;
        __builtin_nontemporal_store(v, reinterpret_cast<u64x2_s *>(p));
    }

    __device__ __forceinline__ u64 mulhi(u64 a, u64 b)
    {
        return __umul64hi(a, b);
    }

    // x*y - floor(x*y'/2^64)*p  (mod 2^64); in [0, 2p) for any 64-bit x when y < p  (ntt.cpp:230-234)
    __device__ __forceinline__ u64 mulmod_lazy(u64 x, u64 y, u64 yshoup, u64 p)
    {
        u64 q = mulhi(x, yshoup);
        return x * y - q * p;
    }

    // canonical Shoup product (multi_special_primes.cpp:13-19)
    __device__ __forceinline__ u64 mulmod_shoup(u64 x, u64 y, u64 yshoup, u64 p)
    {
        u64 t = mulmod_lazy(x, y, yshoup, p);
        return t >= p ? t - p : t;
    }

    // x - floor(x*rdp/2^64)*p, in [0, 2p)  (ntt.cpp:237-241)
    __device__ __forceinline__ u64 barrett_lazy(u64 x, u64 rdp, u64 p)
    {
        return x - mulhi(x, rdp) * p;
    }

    // uintarithsmallmod.h:140-178, exact
    __device__ __forceinline__ u64 barrett_reduce_128(u64 lo, u64 hi, u64 p, u64 cr0, u64 cr1)
    {
        u64 carry = mulhi(lo, cr0);
        u64 t_lo = lo * cr1, t_hi = mulhi(lo, cr1);
        u64 tmp1 = t_lo + carry;
        u64 tmp3 = t_hi + (tmp1 < t_lo);
        u64 u_lo = hi * cr0, u_hi = mulhi(hi, cr0);
        u64 tmp1b = tmp1 + u_lo;
        u64 carry2 = u_hi + (tmp1b < tmp1);
        u64 q = hi * cr1 + tmp3 + carry2;
        u64 r = lo - q * p;
        return r >= p ? r - p : r;
    }

    // Montgomery reduction of a 128-bit accumulator: returns t == acc * 2^-64 (mod p) with t < acc/2^64 + p.
    // Used for dot products against constants that were pre-multiplied by 2^64 mod p on the host, so the
    // canonical result equals the reference's dot_product_mod / barrett_reduce_128 (exact integer identities);
    // 7 multiplier ops instead of the 24 of the two-word Barrett reduction.
    __device__ __forceinline__ u64 redc128(u64 lo, u64 hi, u64 p, u64 ninv)
    {
        const u64 m = lo * ninv;
        return hi + mulhi(m, p) + (lo != 0);
    }
    // canonical residue of a REDC result; `small`: the host proved t < 2p for this stage
    __device__ __forceinline__ u64 redc_finish(u64 t, u64 p, u64 rdp, bool small)
    {
        if (!small)
            t = t - mulhi(t, rdp) * p; // -> [0, 2p)
        return t >= p ? t - p : t;
    }

    // uintarithsmallmod.h:181-207 (x < 2^63)
    __device__ __forceinline__ u64 barrett_reduce_63(u64 x, u64 p, u64 cr1)
    {
        u64 r = x - mulhi(x, cr1) * p;
        return r >= p ? r - p : r;
    }

    // uintarithsmallmod.h:209-221
    __device__ __forceinline__ u64 mul_mod(u64 a, u64 b, u64 p, u64 cr0, u64 cr1)
    {
        return barrett_reduce_128(a * b, mulhi(a, b), p, cr0, cr1);
    }

    // uintarithsmallmod.h:282-290: (a*b + c) mod p with a 128-bit wrapping add
    __device__ __forceinline__ u64 mul_add_mod(u64 a, u64 b, u64 c, u64 p, u64 cr0, u64 cr1)
    {
        u64 lo = a * b, hi = mulhi(a, b);
        u64 lo2 = lo + c;
        hi += (lo2 < lo);
        return barrett_reduce_128(lo2, hi, p, cr0, cr1);
    }

    // 128-bit accumulate: (lo, hi) += a*b  (uintarith.h:942-958)
    __device__ __forceinline__ void mac128(u64 &lo, u64 &hi, u64 a, u64 b)
    {
        u64 pl = a * b, ph = mulhi(a, b);
        u64 nl = lo + pl;
        hi += ph + (nl < lo);
        lo = nl;
    }

    __device__ __forceinline__ u64 add_mod(u64 a, u64 b, u64 p) // polyarithsmallmod.h:261-299
    {
        u64 s = a + b;
        return s >= p ? s - p : s;
    }
    __device__ __forceinline__ u64 sub_mod(u64 a, u64 b, u64 p) // polyarithsmallmod.h:366-404
    {
        u64 d = a - b;
        return a < b ? d + p : d;
    }
    __device__ __forceinline__ u64 neg_mod(u64 a, u64 p) // uintarithsmallmod.h:51-65
    {
        return a ? p - a : 0;
    }

    // ---------------------------------------------------------------------------------------------
    // Hand-selected instruction sequences for the NTT butterflies. The compiler's expansion of a 64-bit
    // mulhi / mullo spends 4 moves per product on zero-extending 32-bit halves into the 64-bit addend of
    // v_mad_u64_u32 and splits the middle sum; here the middle sum keeps its carry (one v_cndmask) and the
    // low products are chained through the 64-bit accumulator of v_mad_u64_u32, whose upper half is
    // don't-care for them. 16 VALU instructions per lazy forward butterfly instead of 21 (10 multiplier
    // ops in both); bit-identical results (everything is arithmetic mod 2^64). tools/ubench_intmul.hip:
    // 1.69 -> 2.04 T butterflies/s with per-lane twiddles.
    // SU = the second factor is wave-uniform (kept in SGPRs; one scalar operand per VOP3 is allowed).
    using u32 = unsigned;
    template <bool SU>
    __device__ __forceinline__ u64 mad64(u32 a, u32 b, u64 c) // a*b + c  (mod 2^64)
    {
        u64 d, cy;
        if (SU)
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "s"(b), "v"(c));
        else
            asm("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "=s"(cy) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    template <bool SU>
    __device__ __forceinline__ u64 mul64(u32 a, u32 b) // a*b
    {
        u64 d, cy;
        if (SU)
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "s"(b));
        else
            asm("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "=s"(cy) : "v"(a), "v"(b));
        return d;
    }
    // floor(x*s / 2^64), exact. The carry of x0*s1 + (x1*s0 + hi32(x0*s0)) is turned into a register inside
    // the same asm statement (gfx950 needs two wait states between a VALU carry write and its VALU read).
    template <bool SU>
    __device__ __forceinline__ u64 mulhi_c(u64 x, u64 s)
    {
        const u32 x0 = static_cast<u32>(x), x1 = static_cast<u32>(x >> 32);
        const u32 s0 = static_cast<u32>(s), s1 = static_cast<u32>(s >> 32);
        const u32 h = __umulhi(x0, s0);
        const u64 A = mad64<SU>(x1, s0, static_cast<u64>(h));
        u64 B;
        u32 cb;
        if (SU)
            asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, vcc"
                : "=&v"(B), "=&v"(cb)
                : "v"(x0), "s"(s1), "v"(A)
                : "vcc");
        else
            asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, vcc"
                : "=&v"(B), "=&v"(cb)
                : "v"(x0), "v"(s1), "v"(A)
                : "vcc");
        const u64 addend = static_cast<u64>(static_cast<u32>(B >> 32)) | (static_cast<u64>(cb) << 32);
        return mad64<SU>(x1, s1, addend);
    }
    __device__ __forceinline__ u64 add_hi32(u64 v, u64 e) // v + (e << 32)
    {
        u32 vh;
        asm("v_add_u32 %0, %1, %2" : "=v"(vh) : "v"(static_cast<u32>(v >> 32)), "v"(static_cast<u32>(e)));
        return static_cast<u64>(static_cast<u32>(v)) | (static_cast<u64>(vh) << 32);
    }
    // lo64(acc + x*w + q*np); np (= 2^64 - p) is always wave-uniform
    template <bool WU>
    __device__ __forceinline__ u64 mullo2_acc(u64 acc, u64 x, u64 w, u64 q, u64 np)
    {
        const u32 x0 = static_cast<u32>(x), x1 = static_cast<u32>(x >> 32);
        const u32 w0 = static_cast<u32>(w), w1 = static_cast<u32>(w >> 32);
        const u32 q0 = static_cast<u32>(q), q1 = static_cast<u32>(q >> 32);
        const u32 n0 = static_cast<u32>(np), n1 = static_cast<u32>(np >> 32);
        u64 E = mul64<WU>(x0, w1);
        E = mad64<WU>(x1, w0, E);
        E = mad64<true>(q0, n1, E);
        E = mad64<true>(q1, n0, E);
        u64 V = mad64<WU>(x0, w0, acc);
        V = mad64<true>(q0, n0, V);
        return add_hi32(V, E);
    }
    // lo64(acc + q*np)
    __device__ __forceinline__ u64 mullo1_acc(u64 acc, u64 q, u64 np)
    {
        const u32 q0 = static_cast<u32>(q), q1 = static_cast<u32>(q >> 32);
        const u32 n0 = static_cast<u32>(np), n1 = static_cast<u32>(np >> 32);
        u64 E = mul64<true>(q0, n1);
        E = mad64<true>(q1, n0, E);
        return add_hi32(mad64<true>(q0, n0, acc), E);
    }
    // mulmod_lazy_np with the sequences above
    template <bool WU>
    __device__ __forceinline__ u64 mulmod_lazy_hs(u64 x, u64 y, u64 yshoup, u64 neg_p)
    {
        return mullo2_acc<WU>(0, x, y, mulhi_c<WU>(x, yshoup), neg_p);
    }
    // canonical Shoup product (MulModShoup, multi_special_primes.cpp:13-19) with a wave-uniform constant
    __device__ __forceinline__ u64 mulmod_shoup_hs(u64 x, u64 y, u64 yshoup, u64 p)
    {
        const u64 t = mulmod_lazy_hs<true>(x, y, yshoup, 0 - p);
        return t >= p ? t - p : t;
    }
    // barrett_lazy with the sequences above: x + q*(2^64-p); rdp is wave-uniform
    __device__ __forceinline__ u64 barrett_lazy_hs(u64 x, u64 rdp, u64 neg_p)
    {
        return mullo1_acc(x, mulhi_c<true>(x, rdp), neg_p);
    }
    // x below 2^7 p, p at least 2^45 (bounds::small_quot_admits) -> the representative in [0, 2p) of the same residue class,
    // with the quotient estimated in single precision from the word's upper half: floor(fl(fl(x >> 32) * c)) is
    // floor(x / p) or one less for c = fl((2^32 / p)(1 - 2^-20)) (ntt_bounds.hpp section 6 proves both directions;
    // tests/bounds_check.cpp runs the same IEEE operations on adversarial words). 6 instructions against the 10 of barrett_lazy_hs.
    __device__ __forceinline__ float small_quot_const(u64 p)
    {
        return static_cast<float>(4294967296.0 / static_cast<double>(p) * (1.0 - 0x1p-20));
    }
    __device__ __forceinline__ u64 reduce_small_quot(u64 x, float c, u64 neg_p)
    {
        // v_cvt_f32_u32, v_mul_f32, v_cvt_u32_f32. (The conversion is spelled as the instruction: from `(float)(u32)(x >> 32)`
        // the compiler sometimes builds the generic 64-bit integer -> float sequence, seven instructions, for a value it could
        // see was below 2^32.)
        float hf;
        asm("v_cvt_f32_u32 %0, %1" : "=v"(hf) : "v"(static_cast<u32>(x >> 32)));
        const u32 q = static_cast<u32>(hf * c);
        const u64 v = mad64<true>(q, static_cast<u32>(neg_p), x);                          // x + q * (2^64 - p), low word
        return add_hi32(v, static_cast<u64>(q * static_cast<u32>(neg_p >> 32)));
    }
    // forward lazy butterfly (ntt.cpp:245-252): X = u + v, Y = u - v + 2p with v = y*w - q*p.
    // X falls out of the multiply-accumulate chain (u is its initial accumulator); Y = (2u + 2p) - X.
    // APX: the quotient estimate without hi32(y0 * s0), i.e. floor(y s / 2^64) or one less: one multiplier instruction
    // saved, the product lands in [0, 3p), and the caller passes 3p as `two_p` (Y = u - v + 3p stays non-negative).
    template <bool SU>
    __device__ __forceinline__ u64 mulhi_apx(u64 x, u64 s)
    {
        const u32 x0 = static_cast<u32>(x), x1 = static_cast<u32>(x >> 32);
        const u32 s0 = static_cast<u32>(s), s1 = static_cast<u32>(s >> 32);
        const u64 A = mul64<SU>(x1, s0);
        u64 B;
        u32 cb;
        if (SU)
            asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, vcc"
                : "=&v"(B), "=&v"(cb)
                : "v"(x0), "s"(s1), "v"(A)
                : "vcc");
        else
            asm("v_mad_u64_u32 %0, vcc, %2, %3, %4\n\ts_nop 1\n\tv_cndmask_b32_e64 %1, 0, 1, vcc"
                : "=&v"(B), "=&v"(cb)
                : "v"(x0), "v"(s1), "v"(A)
                : "vcc");
        const u64 addend = static_cast<u64>(static_cast<u32>(B >> 32)) | (static_cast<u64>(cb) << 32);
        return mad64<SU>(x1, s1, addend);
    }
    // APX == 2 (round 4): the quotient from the three partial products that matter, with NO carry to recover:
    //   q = y1*s1 + hi32(y0*s1) + hi32(y1*s0)   >=   floor(y s / 2^64) - 2
    // (the low halves of the two cross products and hi32(y0*s0) are dropped: each loses less than one unit). The Shoup
    // quotient is itself floor(y w / p) or one less, so the product lands in [0, 4p) and the caller passes 4p as the
    // addend (growth 4p per layer: ntt_bounds.hpp section 2).
    // Instruction count: v_mul_hi_u32, v_mad_u64_u32 (y1*s1 + t1), v_mul_hi_u32, v_mad_u64_u32 (t2*1 + .) = 4, against
    // mul, mad + carry, v_cndmask, v_mov, mad = 5 -- PROVIDED the 64-bit addend (t1, 0) costs no move. A 64-bit operand is
    // an even-aligned register pair, and zero-extending a 32-bit result normally costs the v_mov of the upper half. Here the
    // upper halves are written ONCE per phase (ZeroHi::init, opaque to the compiler, which would otherwise fold the zero and
    // re-materialise it per use; two v_mov per phase) and v_mul_hi_u32 writes the lower half of the same pair in place: `with_low` tells the
    // compiler that the upper half is unchanged, and its coalescer keeps the pair where it is (checked in the ISA: no
    // v_mov between the v_mul_hi_u32 and the v_mad_u64_u32 that reads the pair).
    template <int N>
    struct ZeroHi
    {
        u64 z[N];
        __device__ __forceinline__ void init()
        {
#pragma unroll
            for (int j = 0; j < N; j++)
            {
                u32 h;
                asm volatile("v_mov_b32 %0, 0" : "=v"(h));
                z[j] = static_cast<u64>(h) << 32;
            }
        }
    };
    __device__ __forceinline__ u64 with_low(u64 zp, u32 lo)
    {
        return (zp & 0xFFFFFFFF00000000ull) | lo;
    }
    template <bool SU>
    __device__ __forceinline__ u64 mulhi_apx2(u64 x, u64 s, u64 &zp)
    {
        const u32 x0 = static_cast<u32>(x), x1 = static_cast<u32>(x >> 32);
        const u32 s0 = static_cast<u32>(s), s1 = static_cast<u32>(s >> 32);
        zp = with_low(zp, __umulhi(x0, s1));
        u64 d, cy;
        const u64 A = mad64<SU>(x1, s1, zp);
        asm("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(d), "=s"(cy) : "v"(__umulhi(x1, s0)), "v"(A));
        return d;
    }
    template <bool WU, int APX = 0>
    __device__ __forceinline__ void butterfly_fwd_hs(u64 &xu, u64 &xy, u64 w, u64 wshoup, u64 neg_p, u64 two_p)
    {
        const u64 u = xu;
        static_assert(APX != 2, "level 2 takes a zero-high pair: butterfly_fwd_apx2");
        const u64 X = mullo2_acc<WU>(u, xy, w, APX == 1 ? mulhi_apx<WU>(xy, wshoup) : mulhi_c<WU>(xy, wshoup), neg_p);
        xu = X;
        xy = (u << 1) + two_p - X;
    }
    // IL forward lazy butterflies in lock step: the multiply chain of one butterfly is a string of dependent
    // instructions (a wave issues a dependent VALU instruction only every ~9 cycles, an independent one every
    // ~5.7: tools/ubench_latency.hip), and the compiler, which has no latency model for inline asm, emits each
    // chain back to back. The volatile statements below keep program order, so consecutive instructions belong to
    // different butterflies.
    // `cy` is the (unused) carry-out pair of the chain this instruction belongs to. One pair per chain, passed in
    // and out: were every instruction to get the same scratch pair, the compiler's inline-asm hazard handling
    // (it assumes any two neighbouring asm statements that name a common register need a wait state) would put an
    // s_nop between all of them.
    template <bool SU>
    __device__ __forceinline__ u64 mad64v(u32 a, u32 b, u64 c, u64 &cy)
    {
        u64 d;
        if (SU)
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "+s"(cy) : "v"(a), "s"(b), "v"(c));
        else
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4" : "=v"(d), "+s"(cy) : "v"(a), "v"(b), "v"(c));
        return d;
    }
    template <bool SU>
    __device__ __forceinline__ u64 mul64v(u32 a, u32 b, u64 &cy)
    {
        u64 d;
        if (SU)
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "+s"(cy) : "v"(a), "s"(b));
        else
            asm volatile("v_mad_u64_u32 %0, %1, %2, %3, 0" : "=v"(d), "+s"(cy) : "v"(a), "v"(b));
        return d;
    }
    template <bool SU>
    __device__ __forceinline__ u32 mulhi32v(u32 a, u32 b) // hi32(a*b), program-ordered like mad64v
    {
        u32 d;
        if (SU)
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(a), "s"(b));
        else
            asm volatile("v_mul_hi_u32 %0, %1, %2" : "=v"(d) : "v"(a), "v"(b));
        return d;
    }
    template <bool WU, int IL, int APX = 0>
    __device__ __forceinline__ void butterflies_fwd_hs(u64 (&u)[IL], u64 (&y)[IL], const u64 (&w)[IL], const u64 (&ws)[IL],
                                                       u64 neg_p, u64 two_p)
    {
        static_assert(IL >= 2, "the carry of step 3 is read in step 5: at least two other instructions in between");
        const u32 n0 = static_cast<u32>(neg_p), n1 = static_cast<u32>(neg_p >> 32);
        u64 A[IL], B[IL], E[IL], V[IL], q[IL], carry[IL];
        u32 cb[IL];
        u64 cy[IL] = {};
        static_assert(APX != 2, "level 2 takes zero-high pairs: butterflies_fwd_apx2");
#pragma unroll
        for (int j = 0; j < IL; j++) // 1: A = y1*s0 + hi32(y0*s0)   (APX: without the second term, see butterfly_fwd_hs)
            A[j] = APX == 1 ? mul64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j]), cy[j])
                       : mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j]),
                                    static_cast<u64>(__umulhi(static_cast<u32>(y[j]), static_cast<u32>(ws[j]))), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // 2: E = y0*w1
            E[j] = mul64v<WU>(static_cast<u32>(y[j]), static_cast<u32>(w[j] >> 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // 3: B = y0*s1 + A, carry kept in an SGPR pair
        {
            if (WU)
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4"
                             : "=v"(B[j]), "=s"(carry[j])
                             : "v"(static_cast<u32>(y[j])), "s"(static_cast<u32>(ws[j] >> 32)), "v"(A[j]));
            else
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4"
                             : "=v"(B[j]), "=s"(carry[j])
                             : "v"(static_cast<u32>(y[j])), "v"(static_cast<u32>(ws[j] >> 32)), "v"(A[j]));
        }
#pragma unroll
        for (int j = 0; j < IL; j++) // 4: E += y1*w0
            E[j] = mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(w[j]), E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // 5: carry -> register (>= IL + 1 >= 3 instructions after its producer)
            asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(cb[j]) : "s"(carry[j]));
#pragma unroll
        for (int j = 0; j < IL; j++) // 6: V = u + y0*w0
            V[j] = mad64v<WU>(static_cast<u32>(y[j]), static_cast<u32>(w[j]), u[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // 7: q = y1*s1 + (B >> 32) + (carry << 32) = floor(y*ws / 2^64)
            q[j] = mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j] >> 32),
                              static_cast<u64>(static_cast<u32>(B[j] >> 32)) | (static_cast<u64>(cb[j]) << 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // 8..10: + q*(2^64 - p)
            E[j] = mad64v<true>(static_cast<u32>(q[j]), n1, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mad64v<true>(static_cast<u32>(q[j]), n0, V[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j] >> 32), n0, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
        {
            const u64 X = add_hi32(V[j], E[j]);
            y[j] = (u[j] << 1) + two_p - X;
            u[j] = X;
        }
    }
    // level-2 forms of the two functions above (see mulhi_apx2): `zp` are zero-high pairs of the calling phase
    template <bool WU>
    __device__ __forceinline__ void butterfly_fwd_apx2(u64 &xu, u64 &xy, u64 w, u64 wshoup, u64 neg_p, u64 four_p, u64 &zp)
    {
        const u64 u = xu;
        const u64 X = mullo2_acc<WU>(u, xy, w, mulhi_apx2<WU>(xy, wshoup, zp), neg_p);
        xu = X;
        xy = (u << 1) + four_p - X;
    }
    // (two pairs serve the IL = 4 butterflies: slots 0, 1 take them first, slots 2, 3 once those quotients have read them --
    //  four pairs held through a whole round cost four registers in kernels that run at the 128-register cap)
    template <bool WU, int IL>
    __device__ __forceinline__ void butterflies_fwd_apx2(u64 (&u)[IL], u64 (&y)[IL], const u64 (&w)[IL], const u64 (&ws)[IL],
                                                         u64 neg_p, u64 four_p, u64 (&zp)[2])
    {
        static_assert(IL == 4, "two zero-high pairs, each used by two of the four butterflies");
        const u32 n0 = static_cast<u32>(neg_p), n1 = static_cast<u32>(neg_p >> 32);
        u64 A[IL], E[IL], V[IL], q[IL];
        u32 t1[IL], t2[IL];
        u64 cy[IL] = {};
#pragma unroll
        for (int j = 0; j < 2; j++) // t1 = hi32(y0*s1), into the low half of the zero-high pair
            t1[j] = mulhi32v<WU>(static_cast<u32>(y[j]), static_cast<u32>(ws[j] >> 32));
#pragma unroll
        for (int j = 0; j < IL; j++) // E = y0*w1
            E[j] = mul64v<WU>(static_cast<u32>(y[j]), static_cast<u32>(w[j] >> 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // t2 = hi32(y1*s0)
            t2[j] = mulhi32v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j]));
#pragma unroll
        for (int j = 0; j < 2; j++) // A = y1*s1 + t1
        {
            zp[j] = with_low(zp[j], t1[j]);
            A[j] = mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j] >> 32), zp[j], cy[j]);
        }
#pragma unroll
        for (int j = 2; j < 4; j++)
            t1[j] = mulhi32v<WU>(static_cast<u32>(y[j]), static_cast<u32>(ws[j] >> 32));
#pragma unroll
        for (int j = 0; j < IL; j++) // E += y1*w0
            E[j] = mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(w[j]), E[j], cy[j]);
#pragma unroll
        for (int j = 2; j < 4; j++)
        {
            zp[j - 2] = with_low(zp[j - 2], t1[j]);
            A[j] = mad64v<WU>(static_cast<u32>(y[j] >> 32), static_cast<u32>(ws[j] >> 32), zp[j - 2], cy[j]);
        }
#pragma unroll
        for (int j = 0; j < IL; j++) // V = u + y0*w0
            V[j] = mad64v<WU>(static_cast<u32>(y[j]), static_cast<u32>(w[j]), u[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++) // q = A + t2  (t2 * 1 + A: a 32-bit addend needs no zero-extended pair this way)
            asm volatile("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(q[j]), "+s"(cy[j]) : "v"(t2[j]), "v"(A[j]));
#pragma unroll
        for (int j = 0; j < IL; j++) // + q*(2^64 - p)
            E[j] = mad64v<true>(static_cast<u32>(q[j]), n1, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mad64v<true>(static_cast<u32>(q[j]), n0, V[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j] >> 32), n0, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
        {
            const u64 X = add_hi32(V[j], E[j]);
            y[j] = (u[j] << 1) + four_p - X;
            u[j] = X;
        }
    }
    // lazy product with the level-2 quotient: x*y - q*p in [0, 4p) (any 64-bit x, y < p)
    template <bool WU>
    __device__ __forceinline__ u64 mulmod_lazy_apx2(u64 x, u64 y, u64 yshoup, u64 neg_p, u64 &zp)
    {
        return mullo2_acc<WU>(0, x, y, mulhi_apx2<WU>(x, yshoup, zp), neg_p);
    }
    // ---------------------------------------------------------------------------------------------
    // Carry-free 128-bit dot product  sum_i t_i * c_i  for operands below 2^61 (SEAL_MOD_BIT_COUNT_MAX,
    // util/defines.h:33; checked when the context is built). The reference accumulates 128-bit products lazily
    // (multiply_accumulate_uint64, uintarith.h:942-958) and reduces once; the integer sum is what matters, not
    // how it is accumulated. A 64x64 multiply-accumulate into (lo, hi) costs the compiler ~20 VALU instructions
    // (4 multiplies, carry chains, register moves). Here every partial product goes to a 64-bit accumulator that
    // cannot overflow, each with ONE v_mad_u64_u32 and no carry handling:
    //   t = t1*2^32 + t0 (t1 < 2^29),  t0 = t01*2^16 + t00,  c = c1*2^32 + c0 (c1 < 2^29)
    //   L0 += t00*c0 (< 2^48)   L1 += t01*c0 (< 2^48)   H += t1*c1 (< 2^58)
    //   M[.] += t0*c1, t1*c0 (< 2^61 each; at most 8 products per accumulator)
    // and sum = L0 + L1*2^16 + (M...)*2^32 + H*2^64 is assembled once per dot product.
    struct SplitT // the per-lane factor, split once and reused by every dot product it takes part in
    {
        u32 t0, t1, t00, t01;
        __device__ __forceinline__ SplitT() : t0(0), t1(0), t00(0), t01(0) {}
        __device__ __forceinline__ explicit SplitT(u64 t)
            : t0(static_cast<u32>(t)), t1(static_cast<u32>(t >> 32)), t00(static_cast<u32>(t) & 0xFFFFu),
              t01(static_cast<u32>(t) >> 16)
        {}
    };
    template <int NTERMS>
    struct DotAcc
    {
        static_assert(bounds::dotacc_ok(NTERMS, bounds::kDotAccOperandBits),
                      "carry-free accumulators: NTERMS products of 61-bit operands must fit (ntt_bounds.hpp section 5)");
        static constexpr int NM = (2 * NTERMS + 7) / 8;
        u64 l0 = 0, l1 = 0, h = 0;
        u64 m[NM] = {};
        // c is the wave-uniform constant (its halves are used whole: one scalar operand per v_mad_u64_u32)
        template <int IDX>
        __device__ __forceinline__ void add(const SplitT &t, u64 c)
        {
            static_assert(IDX >= 0 && IDX < NTERMS, "term index");
            const u32 c0 = static_cast<u32>(c), c1 = static_cast<u32>(c >> 32);
            l0 += static_cast<u64>(t.t00) * c0;
            l1 += static_cast<u64>(t.t01) * c0;
            m[(2 * IDX) % NM] += static_cast<u64>(t.t0) * c1;
            m[(2 * IDX + 1) % NM] += static_cast<u64>(t.t1) * c0;
            h += static_cast<u64>(t.t1) * c1;
        }
        __device__ __forceinline__ void finish(u64 &lo, u64 &hi) const
        {
            typedef unsigned __int128 u128;
            u128 acc = (static_cast<u128>(h) << 64) + l0 + (static_cast<u128>(l1) << 16);
#pragma unroll
            for (int i = 0; i < NM; i++)
                acc += static_cast<u128>(m[i]) << 32;
            lo = static_cast<u64>(acc);
            hi = static_cast<u64>(acc >> 64);
        }
    };
    // IL inverse (Gentleman-Sande) lazy butterflies in lock step (BackwardLazy, ntt.cpp:265-272):
    // x' = u + v - (2p if >= 2p), y' = (u - v + 2p) * w lazily. Same instruction discipline as butterflies_fwd_hs.
    // MODE 0: the reference's sequence (addend = 2p). MODE 1 / 2 / 3 (the caller allows any representative and has checked
    // that nothing can wrap): the sum is left unreduced / reduced with barrett_lazy / reduced with the single-precision
    // quotient estimate to [0, 2p), and `addend` is a multiple of p not below the largest second operand, so that
    // u - v + addend stays non-negative.
    template <bool WU, int IL, int MODE = 0>
    __device__ __forceinline__ void butterflies_inv_hs(u64 (&u)[IL], u64 (&y)[IL], const u64 (&w)[IL], const u64 (&ws)[IL],
                                                       u64 neg_p, u64 two_p, u64 rdp = 0)
    {
        static_assert(IL >= 2, "the carry of step 3 is read in step 5: at least two other instructions in between");
        const u32 n0 = static_cast<u32>(neg_p), n1 = static_cast<u32>(neg_p >> 32);
        u64 dlt[IL], A[IL], B[IL], E[IL], V[IL], q[IL], carry[IL];
        u32 cb[IL];
        u64 cy[IL] = {};
#pragma unroll
        for (int j = 0; j < IL; j++)
        {
            dlt[j] = u[j] - y[j] + two_p;
            u64 tt = u[j] + y[j];
            if (MODE == 0)
                u[j] = tt >= two_p ? tt - two_p : tt;
            else if (MODE == 1)
                u[j] = tt;
            else if (MODE == 3) // dense lazy schedule: sums below 16p, the quotient estimated in single precision (rdp = its constant's bits)
                u[j] = reduce_small_quot(tt, __uint_as_float(static_cast<unsigned>(rdp)), neg_p);
            else
                u[j] = barrett_lazy_hs(tt, rdp, neg_p);
        }
#pragma unroll
        for (int j = 0; j < IL; j++)
            A[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(ws[j]),
                              static_cast<u64>(__umulhi(static_cast<u32>(dlt[j]), static_cast<u32>(ws[j]))), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mul64v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(w[j] >> 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
        {
            if (WU)
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4"
                             : "=v"(B[j]), "=s"(carry[j])
                             : "v"(static_cast<u32>(dlt[j])), "s"(static_cast<u32>(ws[j] >> 32)), "v"(A[j]));
            else
                asm volatile("v_mad_u64_u32 %0, %1, %2, %3, %4"
                             : "=v"(B[j]), "=s"(carry[j])
                             : "v"(static_cast<u32>(dlt[j])), "v"(static_cast<u32>(ws[j] >> 32)), "v"(A[j]));
        }
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(w[j]), E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            asm volatile("v_cndmask_b32_e64 %0, 0, 1, %1" : "=v"(cb[j]) : "s"(carry[j]));
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mul64v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(w[j]), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            q[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(ws[j] >> 32),
                              static_cast<u64>(static_cast<u32>(B[j] >> 32)) | (static_cast<u64>(cb[j]) << 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j]), n1, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mad64v<true>(static_cast<u32>(q[j]), n0, V[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j] >> 32), n0, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            y[j] = add_hi32(V[j], E[j]);
    }
    // MODE 1 (lazy sum) of butterflies_inv_hs with the level-2 quotient (mulhi_apx2): x' = u + y unreduced,
    // y' = (u - y + addend) * w in [0, 4p). Only where the schedule's bound on the layer's outputs is the SUM's (at least
    // 4p: every MODE 1 layer, ntt_bounds.hpp section 1), so the schedule and its admission predicate are unchanged.
    template <bool WU, int IL>
    __device__ __forceinline__ void butterflies_inv_apx2(u64 (&u)[IL], u64 (&y)[IL], const u64 (&w)[IL], const u64 (&ws)[IL],
                                                         u64 neg_p, u64 addend, u64 (&zp)[2])
    {
        static_assert(IL == 4, "two zero-high pairs, each used by two of the four butterflies");
        const u32 n0 = static_cast<u32>(neg_p), n1 = static_cast<u32>(neg_p >> 32);
        u64 dlt[IL], A[IL], E[IL], V[IL], q[IL];
        u32 t1[IL], t2[IL];
        u64 cy[IL] = {};
#pragma unroll
        for (int j = 0; j < IL; j++)
        {
            dlt[j] = u[j] - y[j] + addend;
            u[j] = u[j] + y[j];
        }
#pragma unroll
        for (int j = 0; j < 2; j++)
            t1[j] = mulhi32v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(ws[j] >> 32));
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mul64v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(w[j] >> 32), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            t2[j] = mulhi32v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(ws[j]));
#pragma unroll
        for (int j = 0; j < 2; j++)
        {
            zp[j] = with_low(zp[j], t1[j]);
            A[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(ws[j] >> 32), zp[j], cy[j]);
        }
#pragma unroll
        for (int j = 2; j < 4; j++)
            t1[j] = mulhi32v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(ws[j] >> 32));
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(w[j]), E[j], cy[j]);
#pragma unroll
        for (int j = 2; j < 4; j++)
        {
            zp[j - 2] = with_low(zp[j - 2], t1[j]);
            A[j] = mad64v<WU>(static_cast<u32>(dlt[j] >> 32), static_cast<u32>(ws[j] >> 32), zp[j - 2], cy[j]);
        }
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mul64v<WU>(static_cast<u32>(dlt[j]), static_cast<u32>(w[j]), cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            asm volatile("v_mad_u64_u32 %0, %1, %2, 1, %3" : "=v"(q[j]), "+s"(cy[j]) : "v"(t2[j]), "v"(A[j]));
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j]), n1, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            V[j] = mad64v<true>(static_cast<u32>(q[j]), n0, V[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            E[j] = mad64v<true>(static_cast<u32>(q[j] >> 32), n0, E[j], cy[j]);
#pragma unroll
        for (int j = 0; j < IL; j++)
            y[j] = add_hi32(V[j], E[j]);
    }
} // namespace sealhip
