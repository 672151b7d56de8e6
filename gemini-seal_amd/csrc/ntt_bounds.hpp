// ntt_bounds.hpp -- every "this shortcut cannot overflow" argument of the engine in ONE place, as code.
//
// The reference's invariant is that every intermediate of its lazy arithmetic fits a 64-bit word for moduli up to
// SEAL_MOD_BIT_COUNT_MAX = 61 bits (native/src/seal/util/defines.h:33,52-53; butterflies util/ntt.cpp:245-281). The engine
// takes shortcuts that are only bit-exact while *their* intermediates fit too (64-bit words for the integer kernels, the
// 53-bit significand for the FP64 ones). Each shortcut has
//   * its reduction SCHEDULE (which layer reduces what) -- the kernels in ntt.hip read it from here, nowhere else;
//   * a worst-case magnitude RECURRENCE that walks that schedule layer by layer;
//   * an ADMISSION predicate for the launchers, defined as "the recurrence stays below the limit for the largest prime of
//     that size at every supported ring size" -- a constant derived at compile time, not a hand-written inequality.
// tests/bounds_check.cpp enumerates prime sizes 20..61 bits x log n 14..16 x every schedule: admitted => peak below the
// limit, one bit more => the recurrence really overflows (the predicates are tight, not just safe); and a model of the
// floating-point modular product checks the per-product bound used here against exact integer arithmetic on
// adversarial operands. (Round 2 shipped a whole-row inverse with the half-row shape's bound for 24 minutes; with this
// header the same mistake fails a static_assert.)
//
// Plain C++17, no HIP: included by ntt.hip / pipeline.cpp and compiled on its own by the CPU test.
#pragma once
#include <cstdint>

namespace sealhip
{
    namespace bounds
    {
        using u64 = unsigned long long;
        using u128 = unsigned __int128;

        constexpr int kMinHalfLogn = 14, kMaxHalfLogn = 16; // ring sizes served by the single-pass (half-row) kernels
        constexpr int kMaxPrimeBits = 61;                   // SEAL_MOD_BIT_COUNT_MAX (util/defines.h:33)
        constexpr u128 kWord = static_cast<u128>(1) << 64;  // integer kernels: every true value must stay below 2^64
        constexpr long double kFpLimit = 9007199254740992.0L; // 2^53: FP64 kernels hold integers exactly below it

        constexpr u64 max_prime_of_bits(int bits) // the worst case of a predicate "p < 2^bits"
        {
            return (bits >= 64 ? ~u64(0) : (u64(1) << bits) - 1);
        }
        // ... and where a handful of units matter (the FP64 inverse runs to 8p + O(1) against 2^53): the largest value an
        // NTT prime of that size can have, p = 1 (mod 2N) (the context refuses anything else: ntt.cpp:37-52)
        constexpr u64 max_ntt_prime_of_bits(int bits, int logn)
        {
            return bits > logn + 1 ? (u64(1) << bits) - (u64(1) << (logn + 1)) + 1 : 0;
        }

        // =====================================================================================================
        // 1. Inverse transform, integer, lazy sums (ntt_inv_half_kernel<.., LZ = 1>, butterflies_inv_hs<MODE>).
        // A shape with T on-chip layers (T = log n - 1 for the half-row form whose top layer is left to a consumer or to
        // ntt_inv_top_kernel; T = log n for the whole-row form, whose layer T - 1 is BackwardLazyLast with n^-1 folded in).
        // Layer l reduces its sum with barrett_lazy (MODE 2, -> [0, 2p)) only for l == r1 and l == T - 1; elsewhere the sum
        // is left as it is (MODE 1). Values entering layer l are below 2^shift(l) * p; the difference operand gets exactly
        // that bound added so that it stays non-negative. The product is a Shoup product of a 64-bit word: below 2p with the
        // exact quotient (the two reducing layers; in the dense schedule also the first f = log n - 13 layers, whose plain
        // instances would spill with the zero-high pairs live), below kInvLazyProductMult p = 4p with the level-2 quotient
        // (section 2: quotient_shortfall) that every other MODE 1 layer uses since round 4. A MODE 1 layer's outputs are bounded by its unreduced sum, 2^(shift + 1) p >= 4p, so the
        // recurrence below takes the maximum of the two and finds the schedule unchanged.
        constexpr int kInvLazyProductMult = 4;
        // Two schedules (`sched`):
        //   0  SPARSE (rounds 1-3): two reducing layers, r1 and T - 1; for primes with head-room (2^55 / 2^56 per shape);
        //   1  DENSE (round 4): every third layer (l % 3 == 2) and the last reduce, so values never pass 16p: for primes up to
        //      2^60 -- the 60-bit Bsk rows of a BFV multiply, and ciphertext primes of 56 to 60 bits -- which ran the
        //      reference's own sequence (a conditional subtraction in every butterfly) for want of head-room. The reducing
        //      layers bring their SUM below 2p with the single-precision quotient estimate (section 6: quotient below 16) and
        //      form their product with the exact quotient (below 2p); the two layers in between leave sums of up to 4p / 8p
        //      entering and products below 4p (level-2 quotient).
        constexpr int inv_lazy_r1(int T)
        {
            return (T - 1) / 2;
        }
        constexpr int inv_lazy_mode(int T, int l, int sched = 0)
        {
            if (sched == 1)
                return (l % 3 == 2 || l == T - 1) ? 2 : 1;
            return (l == inv_lazy_r1(T) || l == T - 1) ? 2 : 1;
        }
        constexpr int inv_lazy_shift(int T, int l, int sched = 0)
        {
            if (sched == 1)
                return 1 + l % 3; // entering: 2p after a reducing layer (and at the start), then 4p, then 8p
            return 1 + (l <= inv_lazy_r1(T) ? l : l - inv_lazy_r1(T) - 1);
        }
        // worst-case recurrence: the largest true value any instruction of the schedule forms (sum u + y, difference
        // u - y + addend), for inputs below 2p. Also checks the schedule's own claim "values entering layer l are below
        // 2^shift(l) p" -- returns 2^64 (overflow) if the claim were violated, so a wrong shift table cannot pass.
        constexpr u128 inv_lazy_peak(int T, u64 p, int sched = 0)
        {
            u128 bound = static_cast<u128>(2) * p; // inputs: what the reference requires of an inverse transform's input
            u128 peak = bound;
            for (int l = 0; l < T; l++)
            {
                const u128 claimed = static_cast<u128>(p) << inv_lazy_shift(T, l, sched);
                if (bound > claimed)
                    return kWord; // schedule inconsistent
                const u128 sum = 2 * claimed;       // u + y < 2 * claimed
                const u128 diff = 2 * claimed;      // u - y + claimed < 2 * claimed
                peak = sum > peak ? sum : peak;
                peak = diff > peak ? diff : peak;
                // outputs: a reducing layer (MODE 2) leaves sum and product (exact quotient) below 2p; a MODE 1 layer keeps
                // the sum and may leave its product anywhere below 4p (level-2 quotient)
                const u128 prod = static_cast<u128>(kInvLazyProductMult) * p;
                bound = inv_lazy_mode(T, l, sched) == 2 ? static_cast<u128>(2) * p : (sum > prod ? sum : prod);
            }
            return peak;
        }
        constexpr int inv_lazy_max_shift(int T, int sched = 0)
        {
            int m = 0;
            for (int l = 0; l < T; l++)
                m = inv_lazy_shift(T, l, sched) > m ? inv_lazy_shift(T, l, sched) : m;
            return m;
        }
        // admission: largest prime size (bits) for which the recurrence stays below 2^64
        constexpr int inv_lazy_prime_bits(int T, int sched = 0)
        {
            int bits = 0;
            for (int b = 1; b <= kMaxPrimeBits; b++)
                if (inv_lazy_peak(T, max_prime_of_bits(b), sched) < kWord)
                    bits = b;
            return bits;
        }
        constexpr bool inv_lazy_admits(int T, u64 p)
        {
            return p <= max_prime_of_bits(inv_lazy_prime_bits(T));
        }
        constexpr u64 kSmallQuotMinPrime = u64(1) << 45; // section 6
        // the dense schedule: sums up to 16p must fit the word (primes below 2^60) and its reducing layers estimate a quotient
        // below 16 in single precision (section 6: p at least 2^45; the predicate calls small_quot_admits itself)
        constexpr bool small_quot_admits(u64 p, int mult);
        constexpr int kInvDenseSumMult = 16;
        constexpr bool inv_dense_admits(int T, u64 p)
        {
            return T >= 3 && p <= max_prime_of_bits(inv_lazy_prime_bits(T, 1)) && small_quot_admits(p, kInvDenseSumMult);
        }
        static_assert(inv_lazy_prime_bits(13, 1) == 60 && inv_lazy_prime_bits(14, 1) == 60 && inv_lazy_prime_bits(15, 1) == 60 &&
                          inv_lazy_prime_bits(16, 1) == 60,
                      "dense lazy inverse: primes below 2^60 in every shape");
        static_assert(inv_lazy_max_shift(16, 1) == 3, "dense lazy inverse: values entering a layer are below 8p");

        // =====================================================================================================
        // 2. Forward transform, integer (ntt_fwd_half_kernel<.., STRICT = 0 / 2>). The fork's forward butterfly
        // (ForwardLazy, ntt.cpp:245-252) never reduces its first operand: u' = u + v, y' = u - v + 2p with v a lazy Shoup
        // product below 2p, so values grow by 2p per layer; only the last layer (ForwardLazyLast, :254-261) brings its first
        // operand below 2p with a Barrett step. On the 60-bit Bsk rows that growth wraps mod 2^64 and the wrapped words
        // ARE the reference's result: the exact instances have no predicate. Shortcuts that change the
        // representative (only where the consumer reduces whatever it reads):
        //   kNttAnyRep   the last layer keeps its first operand unreduced (fin & 2);
        //   kNttApprox   approximate Shoup quotient (STRICT = 2): the product v lands below g p and y' = u - v + g p, so the
        //                values grow by g p per layer. g = kFwdApxGrowth: 3 for round 2's form (hi32(y0 s0) dropped from the
        //                exact quotient: one unit short at most), 4 for round 4's carry-free form (devmath.hpp mulhi_apx2:
        //                q = y1 s1 + hi32(y0 s1) + hi32(y1 s0) drops the low halves of both cross products as well, three
        //                units short of floor(y w / p) at most, quotient_shortfall below);
        //   unreduced gathered inputs (key-switch mod-up without the conditional subtraction): inputs below 2p not p.
        // All log n layers count (the top one is applied on load).
#ifndef SEALHIP_NTT_APX
#define SEALHIP_NTT_APX 2
#endif
        constexpr int kFwdApxLevel = SEALHIP_NTT_APX;
        static_assert(kFwdApxLevel == 1 || kFwdApxLevel == 2, "approximate-quotient form");
        // how far below floor(y w / p) the quotient estimate can fall, as the number of dropped terms that each lose less
        // than one unit: Shoup's floor(w 2^64 / p) itself (1), hi32(y0 s0) (level 1 and 2: together with the floor of the
        // sum of the cross products, 1), and at level 2 the low halves of the two cross products (1 more: the three dropped
        // pieces sum to less than 3 * 2^32, i.e. lose at most 2 units of 2^32 after the floor).
        constexpr int quotient_shortfall(int level)
        {
            return level == 0 ? 1 : (level == 1 ? 2 : 3);
        }
        constexpr int kFwdApxGrowth = quotient_shortfall(kFwdApxLevel) + 1; // v = y w - q p < (shortfall + 1) p
        constexpr u128 fwd_int_peak(int logn, u64 p, int in_mult, bool apx, bool skip_last_barrett)
        {
            const u128 g = static_cast<u128>(apx ? kFwdApxGrowth : 2) * p; // growth per layer = bound of the product
            u128 bound = static_cast<u128>(in_mult) * p, peak = bound;
            for (int l = 0; l < logn; l++)
            {
                u128 u = bound;
                if (l == logn - 1 && !skip_last_barrett)
                    u = static_cast<u128>(2) * p; // barrett_lazy of the first operand
                const u128 out = u + g;           // u + v < u + g;  u - v + g <= u + g
                peak = out > peak ? out : peak;
                bound = out;
            }
            return peak;
        }
        // one predicate for both shortcuts together with unreduced inputs (the key-switch digit launches use all three),
        // per ring size (the growth is per layer)
        constexpr int fwd_lazy_prime_bits(int logn)
        {
            int bits = 0;
            for (int b = 1; b <= kMaxPrimeBits; b++)
                if (fwd_int_peak(logn, max_prime_of_bits(b), 2, true, true) < kWord)
                    bits = b;
            return bits;
        }
        static_assert(kFwdApxLevel != 1 || (fwd_lazy_prime_bits(14) == 58 && fwd_lazy_prime_bits(16) == 58),
                      "level 1: kNttAnyRep / kNttApprox / unreduced mod-up: primes below 2^58 (50p at log n = 16)");
        static_assert(kFwdApxLevel != 2 || (fwd_lazy_prime_bits(14) == 58 && fwd_lazy_prime_bits(15) == 58 && fwd_lazy_prime_bits(16) == 57),
                      "level 2: 58p / 62p below 2^64 for primes below 2^58 at log n = 14 / 15; 66p at log n = 16: below 2^57");
        constexpr bool fwd_lazy_admits(u64 p, int logn) // kNttAnyRep, kNttApprox, mod-up without the conditional subtraction
        {
            return logn >= kMinHalfLogn && logn <= kMaxHalfLogn && p <= max_prime_of_bits(fwd_lazy_prime_bits(logn));
        }
        // The canonicalising public entry (sealhip_ntt_negacyclic_harvey) documents operands below 4p (include/sealhip.h: what
        // the reference's butterflies are written for), and it is handed caller-owned words: its cheap schedule (approximate
        // quotient, no Barrett step in the last layer, one reduction in the store) is admitted on THAT range, not on the
        // [0, 2p) of the internal launches: inputs below kCanonInMult p.
        constexpr int kCanonInMult = 4;
        constexpr int fwd_canon_output_mult(int logn)
        {
            return kCanonInMult + kFwdApxGrowth * logn;
        }
        constexpr bool fwd_canon_admits(u64 p, int logn)
        {
            return logn >= kMinHalfLogn && logn <= kMaxHalfLogn && fwd_lazy_admits(p, logn) &&
                   fwd_int_peak(logn, p, kCanonInMult, true, true) < kWord;
        }
        // largest output of an approximate-quotient launch, in units of p (documentation + test): 2 + g log n with
        // kNttAnyRep (50 / 66 at log n = 16), else 2 + g (the last layer's first operand is brought below 2p)
        constexpr int fwd_apx_output_mult(int logn, bool anyrep)
        {
            return anyrep ? 2 + kFwdApxGrowth * logn : 2 + kFwdApxGrowth;
        }

        // ---- 2b. Dense lazy FORWARD schedule (round 4; STRICT mode's 56-60-bit rows, ntt_fwd_half_kernel<.., STRICT = 4>).
        // STRICT means "nothing wraps" (SURVEY B.6); Harvey's corrected butterflies guarantee it for any prime with a
        // conditional subtraction per butterfly. Where the consumer takes any representative below 2p (the Bsk rows of the BFV
        // multiply: the tensor product reduces whatever it reads) the same residues come out of the reference's own
        // butterfly -- exact quotient, product below 2p, first operand left as it is: values grow by 2p per layer -- if all 32
        // words of a lane are brought back below 2p often enough that nothing reaches 2^64: with the single-precision quotient
        // estimate (section 6) before rounds 2 and 3 and in the store. Inputs below kFwdDenseInMult p (the range the entry
        // documents), top layer +2p, round 1 +8p = 14p; rounds 2 and 3 from 2p to 10p; the final round's f = log n - 13 layers
        // from 10p to at most 16p: every value is BELOW fwd_dense_peak_mult(log n) p <= 16p < 2^64 for p < 2^60.
        constexpr int kFwdDenseInMult = 4;
        constexpr bool fwd_dense_reduce_before_round(int r) // r = 1..3 register rounds, 4 = the final round
        {
            return r == 2 || r == 3;
        }
        constexpr int fwd_dense_peak_mult(int logn)
        {
            int b = kFwdDenseInMult + 2, peak = b; // after the top layer
            for (int r = 1; r <= 4; r++)
            {
                if (fwd_dense_reduce_before_round(r))
                    b = 2;
                const int layers = r <= 3 ? 4 : logn - 13;
                for (int l = 0; l < layers; l++)
                    b += 2;
                peak = b > peak ? b : peak;
            }
            return peak; // (also the bound of the stored words before the store's reduction)
        }
        constexpr bool fwd_dense_admits(u64 p, int logn)
        {
            // (p below 2^60 in every shape although 14p would allow a little more: the stored words, below 2p, are operands of
            //  the carry-free dot products of section 3, which take words below 2^61)
            return logn >= kMinHalfLogn && logn <= kMaxHalfLogn && p < (u64(1) << 60) &&
                   small_quot_admits(p, fwd_dense_peak_mult(logn)) && static_cast<u128>(fwd_dense_peak_mult(logn)) * p <= kWord;
        }
        static_assert(fwd_dense_peak_mult(14) == 14 && fwd_dense_peak_mult(15) == 14 && fwd_dense_peak_mult(16) == 16,
                      "dense forward schedule: 14p after round 1, at most 16p after the final round at N = 2^16");

        // =====================================================================================================
        // 3. Tensor product formed by the inverse transform's load (dyadic_redc in ntt.hip): t = (sum of NP products) *
        // 2^-64 mod p as ONE Montgomery reduction, t < sum / 2^64 + p. Carry-free accumulation needs operands below 2^61
        // (upper halves below 2^29: the middle accumulator holds 2 NP products below 2^61 each, NP <= 2). For the inverse
        // transform's inputs t must be below 2p: sum / 2^64 <= p.
        constexpr bool tensor_redc_ok(u64 p, int operand_mult, int np)
        {
            const u128 ob = static_cast<u128>(operand_mult) * p; // operands below operand_mult * p, i.e. at most ob - 1
            if (ob > (static_cast<u128>(1) << 61) || np > 2)
                return false;
            // sum <= np (ob - 1)^2 < 2^123; t < sum / 2^64 + p, so t < 2p as soon as sum <= p 2^64 (< 2^125): exact in u128
            return (ob - 1) * (ob - 1) * np <= static_cast<u128>(p) * kWord;
        }
        // operands below 4p (what the lazy forward transform stores) for primes below 2^59 ...
        constexpr int tensor_prime_bits(int operand_mult)
        {
            int bits = 0;
            for (int b = 1; b <= kMaxPrimeBits; b++)
                if (tensor_redc_ok(max_prime_of_bits(b), operand_mult, 2))
                    bits = b;
            return bits;
        }
        constexpr int kTensorPrimeBits4p = tensor_prime_bits(4);
        constexpr int kTensorPrimeBits2p = tensor_prime_bits(2);
        static_assert(kTensorPrimeBits4p == 59, "fused tensor product on [0, 4p) operands: ciphertext primes below 2^59");
        static_assert(kTensorPrimeBits2p == 60, "... on [0, 2p) operands (kNttReduceOut rows): the 60-bit Bsk primes");
        constexpr bool tensor_admits_4p(u64 p)
        {
            return p <= max_prime_of_bits(kTensorPrimeBits4p);
        }
        constexpr bool tensor_admits_2p(u64 p)
        {
            return p <= max_prime_of_bits(kTensorPrimeBits2p);
        }
        // ... and on what an approximate-quotient forward launch WITHOUT kNttAnyRep stores (op_bfv_multiply's q rows when the
        // tensor product is fused: the last layer keeps its Barrett step, outputs below (2 + g) p, section 2)
        constexpr int kTensorPrimeBitsApx = tensor_prime_bits(fwd_apx_output_mult(kMaxHalfLogn, false));
        static_assert(kTensorPrimeBitsApx == (kFwdApxLevel == 2 ? 57 : 58), "fused tensor product on approximate-quotient rows");
        constexpr bool tensor_admits_apx(u64 p)
        {
            return p <= max_prime_of_bits(kTensorPrimeBitsApx);
        }

        // =====================================================================================================
        // 4. FP64 instances (primes below 2^50; devmath.hpp fp_mulmod / fp_reduce). With u = 2^-53 the unit roundoff:
        //   h = fl(y w), l = y w - h exactly (|l| <= u |y w|); t = fl(h * fl(1/p)) = (h / p)(1 + e), |e| <= 2u + u^2;
        //   q = rint(t) (|q - t| <= 1/2; for |t| >= 2^52 t is an integer already); r = (h - q p) + l.
        //   |h - q p| <= |h| |e| + p / 2 and |l| <= u |h|(1 + u), so
        //       |r| <= p / 2 + 3 u |y| w (1 + 2^-50)  <=  (1/2 + 3 * 2^-53 |y|) p        (+ the tiny slack below)
        // -- NOT (1/2 + 2^-52 |y|) p as round 2 assumed: the exact rounding error l and the rounding of h * (1/p) add up
        // (p = 1125899886395393, y = 3857024279003347, w = 1125899289087767 gives |r| / p = 1.4006 against 1.3564).
        // h - q p and the final sum are integers: exact as long as they stay below 2^53 in magnitude.
        // A forward layer therefore takes a bound B (on every value) to B + (1/2 + 3 * 2^-53 B) p; with p <= U = 2^50 that
        // is 1.375 B + U / 2: from a reduction (B = U / 2) FIVE layers stay below 8 U = 2^53 (1.19, 2.13, 3.43, 5.22,
        // 7.68), the sixth does not (11.06). Round 2's schedule had two spans of six layers.
        constexpr long double kFpSlack = 1.0L + 0x1p-40L;
        constexpr long double fp_mul_bound(long double y, long double p) // |fp_mulmod(y, w, p)| for |y| <= y, 0 <= w < p
        {
            return 0.5L * p + 3.0L * 0x1p-53L * y * p * kFpSlack;
        }
        constexpr long double fp_reduce_bound(long double x, long double p) // |fp_reduce(x)| for |x| <= x < 2^53
        {
            return 0.5L * p + x * 0x1p-51L + 1.0L;
        }
        // Forward schedule (ntt_fwd_half_kernel<.., STRICT = 3>): the top layer on raw inputs (below in_bound: 2^52 for
        // gathered words, p for residues), all values reduced; then the T = log n - 1 on-chip layers i = 0 .. T - 1
        // (rounds 1-3: four each, final round: the rest), all 32 values of a lane reduced BEFORE layers 5 and 10; the store
        // canonicalises (a reduction of values below 2^53).
        constexpr bool fp_fwd_reduce_before_layer(int i)
        {
            return i == 5 || i == 10;
        }
        constexpr long double fp_fwd_peak(int logn, long double p, long double in_bound)
        {
            long double b = in_bound, peak = in_bound;
            // top layer: u + r, u - r with r = fp_mulmod(y, w)
            b = b + fp_mul_bound(b, p);
            peak = b > peak ? b : peak;
            b = fp_reduce_bound(b, p);
            for (int i = 0; i < logn - 1; i++)
            {
                if (fp_fwd_reduce_before_layer(i))
                    b = fp_reduce_bound(b, p);
                b = b + fp_mul_bound(b, p);
                peak = b > peak ? b : peak;
            }
            return peak; // (the canonicalisation reads values up to `peak`)
        }
        // Inverse schedule (ntt_inv_half_kernel<.., LZ = 2>): Gentleman-Sande layers l = 0 .. T - 1 on inputs below 2p,
        // u' = u + y (the bound doubles), y' = fp_mulmod(u - y, w); both outputs reduced after layers l % 4 == 1 except the
        // last on-chip layer (whose outputs the store, or the whole-row form's top layer, takes as they are).
        constexpr bool fp_inv_reduce_after_layer(int T, int l)
        {
            return l % 4 == 1 && l != T - 1;
        }
        constexpr long double fp_inv_peak(int T, long double p)
        {
            long double b = 2.0L * p, peak = b;
            for (int l = 0; l < T; l++)
            {
                const long double s = 2.0L * b; // |u + y|, |u - y| <= 2b
                peak = s > peak ? s : peak;
                const long double r = fp_mul_bound(s, p);
                b = s > r ? s : r;
                if (fp_inv_reduce_after_layer(T, l))
                    b = fp_reduce_bound(b, p);
            }
            return peak;
        }
        constexpr int fp_prime_bits()
        {
            int bits = 0;
            for (int b = kMaxHalfLogn + 2; b <= 52; b++)
            {
                bool ok = true;
                for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
                {
                    const long double p = static_cast<long double>(max_ntt_prime_of_bits(b, logn));
                    ok = ok && fp_fwd_peak(logn, p, 0x1p52L) < kFpLimit;
                    ok = ok && fp_inv_peak(logn - 1, p) < kFpLimit; // half-row form
                    if (logn <= 15)
                        ok = ok && fp_inv_peak(logn, p) < kFpLimit; // whole-row form (N = 2^14, 2^15)
                }
                if (ok)
                    bits = b;
            }
            return bits;
        }
        constexpr int kFpPrimeBits = fp_prime_bits();
        static_assert(kFpPrimeBits == 50, "FP64 instances: primes below 2^50 (and raw inputs below 2^52)");
        constexpr u64 kFpInputBound = u64(1) << 52; // raw words a forward FP64 launch may be handed (fp_from_u64's range too)
        constexpr bool fp_admits(u64 p)
        {
            return p <= max_prime_of_bits(kFpPrimeBits);
        }
        // =====================================================================================================
        // 5. Carry-free 128-bit dot products of the BEHZ kernels (devmath.hpp DotAcc<NTERMS>): sum_i t_i c_i with every
        // partial product in a 64-bit accumulator that is only assembled at the end. With operands below 2^bits
        // (t = t1 2^32 + t0, t0 = t01 2^16 + t00; c = c1 2^32 + c0): l0 += t00 c0 and l1 += t01 c0 (below 2^48 each), h +=
        // t1 c1 (below 2^(2 bits - 64)), and NM = (2 NTERMS + 7) / 8 middle accumulators that take the 2 NTERMS products
        // t0 c1, t1 c0 (below 2^(bits)) round-robin -- at most ceil(2 NTERMS / NM) each. Every accumulator must stay below 2^64.
        constexpr bool dotacc_ok(int nterms, int bits)
        {
            if (nterms < 1 || bits < 33 || bits > 64)
                return false;
            const u128 lim = kWord;
            const int nm = (2 * nterms + 7) / 8;
            const int per_m = (2 * nterms + nm - 1) / nm;
            const u128 low = static_cast<u128>(nterms) << 48;                  // l0, l1
            const u128 mid = static_cast<u128>(per_m) << bits;                 // t0 c1 < 2^32 2^(bits-32)
            const u128 high = static_cast<u128>(nterms) << (2 * (bits - 32));  // t1 c1
            return low <= lim && mid <= lim && high <= lim;
        }
        constexpr int kDotAccOperandBits = 61; // SEAL_MOD_BIT_COUNT_MAX: residues and constants of every prime the context admits
        constexpr int dotacc_max_terms()
        {
            int n = 0;
            for (int t = 1; t <= 4096; t++)
                if (dotacc_ok(t, kDotAccOperandBits))
                    n = t;
                else
                    break;
            return n;
        }
        static_assert(dotacc_max_terms() >= 64, "DotAcc on 61-bit operands: at least 64 terms (the kernels use up to k + 2 <= 34)");

        // =====================================================================================================
        // 6. Small quotients estimated in single precision (devmath.hpp reduce_small_quot): for a word x below M p the
        // quotient estimate is q' = trunc(fl(fl(x >> 32) * c)) with c = fl32(fl64(2^32 / p) (1 - 2^-20)); the reduced value
        // x - q' p must land in [0, 2p), i.e. q' in {floor(x / p) - 1, floor(x / p)}.
        //   * never too large: with e = 2^-24 the unit roundoff, fl(x >> 32) <= (x / 2^32)(1 + e), c <= (2^32 / p)(1 - 2^-20)
        //     (1 + 2^-52)(1 + e), and the product rounds once more: q' <= (x / p)(1 - 2^-20)(1 + e)^3 (1 + 2^-52) < x / p.
        //   * at most one short: fl(x >> 32) >= ((x - 2^32) / 2^32)(1 - e) drops the low half (less than 2^32 / p of
        //     quotient), every rounding loses at most e relatively and the bias 2^-20: x / p - q'real < 2^32 / p +
        //     M (2^-20 + 3 e + 2^-51) =: loss(M, p); trunc loses less than one more. loss < 1 makes q' >= floor(x / p) - 1.
        // Admission: loss(M, p) <= 1/2 (a factor two of margin), which needs p >= 2^33 and M <= 2^18; the launchers ask for
        // M <= 128 and p >= 2^45 (the hi word must also fit single precision's exponent range trivially, and M p < 2^64).
        constexpr bool small_quot_admits(u64 p, int mult)
        {
            if (p < kSmallQuotMinPrime || mult < 1 || mult > 128 || static_cast<u128>(mult) * p > kWord)
                return false;
            const long double loss = 0x1p32L / static_cast<long double>(p) +
                                     static_cast<long double>(mult) * (0x1p-20L + 3 * 0x1p-24L + 0x1p-51L);
            return loss <= 0.5L;
        }
    } // namespace bounds
} // namespace sealhip
