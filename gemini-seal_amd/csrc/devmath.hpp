// devmath.hpp -- 64-bit modular arithmetic for gfx950 device code.
//
// CDNA4 has no 64x64 multiplier: a 64-bit product is built from v_mad_u64_u32 / v_mul_lo_u32.
// The helpers below state the exact functions of the reference's scalar primitives
// (SURVEY Appendix A.1); where the reference returns a canonical residue any exact reduction
// yields identical bits, where it returns a "lazy" representative the operation sequence is kept.
#pragma once

#include <hip/hip_runtime.h>

#include <cstdint>

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
    };

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

    // The same value computed as lo64(x*y + q*(2^64 - p)): the 64-bit subtract (v_sub_co / s_nop / v_subb on
    // gfx950) becomes one v_lshl_add_u64. Identical bits: all arithmetic is mod 2^64. neg_p = 2^64 - p.
    __device__ __forceinline__ u64 mulmod_lazy_np(u64 x, u64 y, u64 yshoup, u64 neg_p)
    {
        const u64 q = mulhi(x, yshoup);
        return x * y + q * neg_p;
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
} // namespace sealhip
