// bounds_check.cpp -- CPU test of gemini-seal_amd/csrc/ntt_bounds.hpp (built and run by tests/test_host.py).
// 1. Enumerates prime sizes 20..61 bits x log n 14..16 x every shortcut schedule: admitted => the worst-case recurrence
//    stays below 2^64 (integer) / 2^53 (FP64); rejected by one bit => the recurrence really overflows (tight).
// 2. Regression for the round-2 bug (whole-row inverse admitted with the half-row shape's bound): fails here.
// 3. Checks the recurrences themselves against executions: a bit-level model of the FP64 modular product (same IEEE
//    operations as devmath.hpp: multiply, fma, rint) on adversarial operands against exact __int128 arithmetic, and whole
//    transforms run through the schedules with every intermediate magnitude tracked -- none may exceed what the
//    recurrence predicts, and every value must stay congruent to an exact integer shadow.
// Invariant restated: native/src/seal/util/defines.h:52-53 (lazy arithmetic fits the word), butterflies util/ntt.cpp:245-281.
#include <cfenv>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <random>
#include <vector>

#include "../gemini-seal_amd/csrc/ntt_bounds.hpp"

using namespace sealhip::bounds;
typedef __int128 i128;

static int failures = 0;
#define CHECK(cond, ...)                       \
    do                                         \
    {                                          \
        if (!(cond))                           \
        {                                      \
            failures++;                        \
            std::printf("FAIL %s:%d: ", __FILE__, __LINE__); \
            std::printf(__VA_ARGS__);          \
            std::printf("\n");                 \
        }                                      \
    } while (0)

// ---- the FP64 modular product exactly as devmath.hpp computes it (IEEE double, round to nearest even)
static double fp_reduce_model(double x, double p, double pinv)
{
    return std::fma(-std::nearbyint(x * pinv), p, x);
}
static double fp_mulmod_model(double y, double w, double p, double pinv)
{
    const double h = y * w;
    const double l = std::fma(y, w, -h);
    const double q = std::nearbyint(h * pinv);
    return std::fma(-q, p, h) + l;
}
static i128 mod_i128(i128 a, i128 p)
{
    i128 r = a % p;
    return r < 0 ? r + p : r;
}

static void enumerate_predicates()
{
    int checked = 0;
    // 1a. inverse, integer lazy sums: every shape the launchers use (T = log n - 1 half-row, T = log n whole-row)
    for (int T = kMinHalfLogn - 1; T <= kMaxHalfLogn; T++)
        for (int bits = 20; bits <= kMaxPrimeBits; bits++)
        {
            const u64 p = max_prime_of_bits(bits);
            const bool adm = inv_lazy_admits(T, p);
            const u128 peak = inv_lazy_peak(T, p);
            if (adm)
                CHECK(peak < kWord, "inverse lazy T=%d bits=%d admitted but the recurrence overflows", T, bits);
            else
                CHECK(peak >= kWord, "inverse lazy T=%d bits=%d rejected although the recurrence fits (predicate not tight)", T, bits);
            CHECK(inv_lazy_prime_bits(T) == 63 - inv_lazy_max_shift(T), "inverse lazy T=%d: closed form 63 - max_shift", T);
            checked++;
        }
    // 1a'. the dense schedule (round 4): admitted up to 60 bits in every shape, tight, and never below 2^45
    for (int T = kMinHalfLogn - 1; T <= kMaxHalfLogn; T++)
        for (int bits = 20; bits <= kMaxPrimeBits; bits++)
        {
            const u64 p = max_prime_of_bits(bits);
            const bool adm = inv_dense_admits(T, p);
            const u128 peak = inv_lazy_peak(T, p, 1);
            if (adm)
                CHECK(peak < kWord && small_quot_admits(p, kInvDenseSumMult) && peak <= static_cast<u128>(kInvDenseSumMult) * p,
                      "dense inverse T=%d bits=%d admitted but the recurrence overflows or passes 16p", T, bits);
            else
                CHECK(peak >= kWord || bits <= 45, "dense inverse T=%d bits=%d rejected although it fits (not tight)", T, bits);
            checked++;
        }
    // 1b. forward integer shortcuts (kNttAnyRep, kNttApprox, unreduced inputs), every combination that is launched
    for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
        for (int bits = 20; bits <= kMaxPrimeBits; bits++)
        {
            const u64 p = max_prime_of_bits(bits);
            for (int in_mult = 1; in_mult <= 2; in_mult++)
                for (int apx = 0; apx <= 1; apx++)
                    for (int skip = 0; skip <= 1; skip++)
                    {
                        if (!apx && !skip && in_mult == 1)
                            continue; // the reference's own sequence: no predicate (its wrap-around is the result)
                        const u128 peak = fwd_int_peak(logn, p, in_mult, apx, skip);
                        if (fwd_lazy_admits(p, logn))
                            CHECK(peak < kWord, "forward logn=%d bits=%d in<%dp apx=%d skip=%d admitted but overflows", logn, bits,
                                  in_mult, apx, skip);
                        checked++;
                    }
        }
    {
        // the canonicalising entry's schedule on the documented input range [0, 4p)
        for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
            for (int bits = 20; bits <= kMaxPrimeBits; bits++)
            {
                const u64 p = max_prime_of_bits(bits);
                if (fwd_canon_admits(p, logn))
                {
                    CHECK(fwd_int_peak(logn, p, kCanonInMult, true, true) < kWord, "canonical forward logn=%d bits=%d admitted but overflows", logn, bits);
                    CHECK(fwd_int_peak(logn, p, kCanonInMult, true, true) <= static_cast<u128>(fwd_canon_output_mult(logn)) * p, "canonical output bound");
                    if (bits >= 46)
                        CHECK(small_quot_admits(p, fwd_canon_output_mult(logn)), "small quotient must cover the canonical store logn=%d bits=%d", logn, bits);
                }
                checked++;
            }
        // tight: one bit above the bound the full combination overflows, at every ring size
        for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
        {
            const int bits = fwd_lazy_prime_bits(logn);
            CHECK(fwd_int_peak(logn, max_prime_of_bits(bits + 1), 2, true, true) >= kWord, "forward lazy predicate is not tight (log n %d)", logn);
            CHECK(!fwd_lazy_admits(max_prime_of_bits(bits + 1), logn) && fwd_lazy_admits(max_prime_of_bits(bits), logn), "forward predicate edge");
            CHECK(fwd_int_peak(logn, max_prime_of_bits(bits), 2, true, true) / max_prime_of_bits(bits) <
                      static_cast<u128>(fwd_apx_output_mult(logn, true)) + 1,
                  "documented output bound (2 + g log n) p");
        }
    }
    // 1c. fused tensor product
    for (int bits = 20; bits <= kMaxPrimeBits; bits++)
    {
        const u64 p = max_prime_of_bits(bits);
        CHECK(tensor_admits_4p(p) == tensor_redc_ok(p, 4, 2), "tensor 4p bits=%d", bits);
        CHECK(tensor_admits_2p(p) == tensor_redc_ok(p, 2, 2), "tensor 2p bits=%d", bits);
        CHECK(tensor_admits_apx(p) == tensor_redc_ok(p, fwd_apx_output_mult(kMaxHalfLogn, false), 2), "tensor on approximate-quotient rows bits=%d", bits);
        if (tensor_admits_4p(p))
            CHECK(tensor_redc_ok(p, 4, 1), "one product must fit where two do");
        checked += 2;
    }
    // 1d. FP64 schedules
    for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
        for (int bits = 20; bits <= 52; bits++)
        {
            const long double p = static_cast<long double>(max_ntt_prime_of_bits(bits, logn));
            const long double f = fp_fwd_peak(logn, p, 0x1p52L), ih = fp_inv_peak(logn - 1, p), iw = fp_inv_peak(logn, p);
            if (fp_admits(max_ntt_prime_of_bits(bits, logn)))
            {
                CHECK(f < kFpLimit, "FP64 forward logn=%d bits=%d admitted but reaches 2^53", logn, bits);
                CHECK(ih < kFpLimit, "FP64 inverse (half) logn=%d bits=%d admitted but reaches 2^53", logn, bits);
                if (logn <= 15)
                    CHECK(iw < kFpLimit, "FP64 inverse (whole) logn=%d bits=%d admitted but reaches 2^53", logn, bits);
            }
            else if (bits == kFpPrimeBits + 1)
                CHECK(f >= kFpLimit && ih >= kFpLimit, "FP64 predicate not tight at %d bits", bits);
            checked += 3;
        }
    // the round-2 schedule (reductions before layers 6 and 12: spans of six) does NOT pass the sound recurrence
    {
        const long double p = static_cast<long double>(max_ntt_prime_of_bits(50, 15));
        long double b = fp_reduce_bound(0x1p52L + fp_mul_bound(0x1p52L, p), p), peak = 0;
        for (int i = 0; i < 14; i++)
        {
            if (i == 6 || i == 12)
                b = fp_reduce_bound(b, p);
            b += fp_mul_bound(b, p);
            peak = b > peak ? b : peak;
        }
        CHECK(peak >= kFpLimit, "the six-layer spans of round 2 should fail the sound bound (peak %.3Lf U)", peak / 0x1p50L);
    }
    std::printf("enumerated %d (schedule, ring, prime size) cases\n", checked);
}

static void regression_whole_row()
{
    // round 2, commit 5a53534: the whole-row inverse of a ring of 2^15 (T = 15 on-chip layers) was admitted with the half-row
    // shape's bound (T = 14). With the predicate taken from the launched instance's layer count this cannot happen; the
    // mistaken pairing itself must be detectably unsafe:
    bool unsafe = false;
    for (int bits = 20; bits <= kMaxPrimeBits; bits++)
    {
        const u64 p = max_prime_of_bits(bits);
        if (inv_lazy_admits(14, p) && inv_lazy_peak(15, p) >= kWord)
            unsafe = true;
    }
    CHECK(unsafe, "half-row predicate applied to the whole-row shape should be caught (56-bit primes at N = 2^15)");
    CHECK(inv_lazy_prime_bits(14) == 56 && inv_lazy_prime_bits(15) == 55, "bounds of the two shapes");
}

static void fp_product_model()
{
    // the advisor's counterexample to the round-2 bound (0.5 + 2^-52 |y|) p
    {
        const double p = 1125899886395393.0, y = 3857024279003347.0, w = 1125899289087767.0;
        const double r = fp_mulmod_model(y, w, p, 1.0 / p);
        const long double ratio = fabsl(static_cast<long double>(r)) / p;
        CHECK(ratio > 0.5L + 0x1p-52L * y, "counterexample should exceed the old bound (ratio %.4Lf)", ratio);
        CHECK(fabsl(static_cast<long double>(r)) <= fp_mul_bound(y, p), "counterexample within the sound bound");
        CHECK(mod_i128(static_cast<i128>(r), static_cast<i128>(p)) ==
                  mod_i128(static_cast<i128>(y) * static_cast<i128>(w), static_cast<i128>(p)),
              "product model is exact");
    }
    std::mt19937_64 rng(12345);
    long double worst = 0;
    const u64 primes[] = { 1125899886395393ull, 1125899903107073ull, 1125899906826241ull, max_ntt_prime_of_bits(50, 14),
                           max_ntt_prime_of_bits(49, 16), 562949953216513ull, 1099511480321ull };
    for (u64 pu : primes)
    {
        const double p = static_cast<double>(pu), pinv = 1.0 / p;
        for (int it = 0; it < 400000; it++)
        {
            // magnitudes up to the 2^53 limit, concentrated near the top; w near p and anywhere
            const int yb = 40 + static_cast<int>(rng() % 13);
            u64 ym = (rng() >> (64 - yb)) | (u64(1) << (yb - 1));
            if (it % 7 == 0)
                ym = (u64(1) << 53) - 1 - (rng() & 0xFFFF);
            const double y = (rng() & 1) ? static_cast<double>(ym) : -static_cast<double>(ym);
            u64 wu = rng() % pu;
            if (it % 5 == 0)
                wu = pu - 1 - (rng() & 0xFFFFF) % pu;
            const double w = static_cast<double>(wu);
            const double r = fp_mulmod_model(y, w, p, pinv);
            const long double ar = fabsl(static_cast<long double>(r)), bd = fp_mul_bound(fabs(y), p);
            if (ar > bd)
            {
                CHECK(false, "|r| = %.1Lf exceeds the bound %.1Lf (p=%llu y=%.0f w=%llu)", ar, bd, pu, y, wu);
                return;
            }
            const long double excess = (ar / p - 0.5L) / (fabsl(static_cast<long double>(y)) * 0x1p-53L);
            worst = excess > worst ? excess : worst;
            if (ar < 0x1p53L)
            {
                const i128 want = mod_i128(static_cast<i128>(static_cast<long long>(y)) * static_cast<i128>(wu), pu);
                if (mod_i128(static_cast<i128>(static_cast<long long>(r)), pu) != want)
                {
                    CHECK(false, "product not exact (p=%llu y=%.0f w=%llu)", pu, y, wu);
                    return;
                }
            }
        }
    }
    std::printf("FP64 product model: worst observed (|r|/p - 1/2) / (2^-53 |y|) = %.3Lf (bound 3)\n", worst);
    CHECK(worst <= 3.0L, "observed coefficient above 3");
    CHECK(worst > 1.0L, "the search should find operands beyond round 2's coefficient 2 * 2^-53 ... (found %.3Lf)", worst);
}

// ---- whole forward transform through the FP64 schedule, magnitudes tracked, exact shadow mod p.
// The modulus need not be prime for this (no inverse is taken): the butterflies are sums and modular products.
static void fp_forward_execution(int logn, u64 pu, int pattern)
{
    const int n = 1 << logn;
    const double p = static_cast<double>(pu), pinv = 1.0 / p;
    std::mt19937_64 rng(logn * 1000 + pattern);
    std::vector<double> x(n);
    std::vector<u64> shadow(n);
    for (int i = 0; i < n; i++)
    {
        u64 v;
        if (pattern == 0)
            v = (u64(1) << 52) - 1; // the largest raw word a gathered launch may see
        else if (pattern == 1)
            v = (i & 1) ? (u64(1) << 52) - 1 : 0;
        else if (pattern == 2)
            v = pu - 1;
        else
            v = rng() & ((u64(1) << 52) - 1);
        x[i] = static_cast<double>(v);
        shadow[i] = v % pu;
    }
    long double peak = 0;
    auto layer = [&](int gap, bool reduce_first) {
        if (reduce_first)
            for (int i = 0; i < n; i++)
                x[i] = fp_reduce_model(x[i], p, pinv);
        for (int blk = 0, t = 0; blk < n; blk += 2 * gap, t++)
        {
            // adversarial "twiddles": near p for one half of the blocks, random for the rest
            const u64 wu = (t & 1) ? pu - 1 - (rng() & 0xFFFF) : rng() % pu;
            const double w = static_cast<double>(wu);
            for (int j = blk; j < blk + gap; j++)
            {
                const double u = x[j], y = x[j + gap];
                const long double ay = fabsl(static_cast<long double>(y));
                peak = ay > peak ? ay : peak;
                const double r = fp_mulmod_model(y, w, p, pinv);
                x[j] = u + r;
                x[j + gap] = u - r;
                const u64 sv = static_cast<u64>(static_cast<u128>(shadow[j + gap]) * wu % pu);
                const u64 su = shadow[j];
                shadow[j] = (su + sv) % pu;
                shadow[j + gap] = (su + pu - sv) % pu;
                const long double a0 = fabsl(static_cast<long double>(x[j])), a1 = fabsl(static_cast<long double>(x[j + gap]));
                peak = a0 > peak ? a0 : peak;
                peak = a1 > peak ? a1 : peak;
            }
        }
    };
    layer(n >> 1, false); // top layer on raw inputs
    for (int i = 0; i < n; i++)
        x[i] = fp_reduce_model(x[i], p, pinv);
    for (int i = 0; i < logn - 1; i++)
        layer(n >> (i + 2), fp_fwd_reduce_before_layer(i));
    const long double predicted = fp_fwd_peak(logn, static_cast<long double>(pu), 0x1p52L);
    CHECK(peak < kFpLimit, "FP64 forward execution logn=%d pattern=%d reached 2^53 (%.3Lf U)", logn, pattern, peak / 0x1p50L);
    CHECK(peak <= predicted, "execution peak %.4Lf U above the recurrence %.4Lf U", peak / 0x1p50L, predicted / 0x1p50L);
    int bad = 0;
    for (int i = 0; i < n; i++)
    {
        double r = fp_reduce_model(x[i], p, pinv);
        if (r < 0)
            r += p;
        bad += static_cast<u64>(r) != shadow[i];
    }
    CHECK(bad == 0, "FP64 forward execution logn=%d pattern=%d: %d words differ from the exact shadow", logn, pattern, bad);
}

// ---- inverse integer lazy-sum schedule executed on 64-bit words with a 128-bit shadow of every true value
static u64 quotient_model(u64 y, u64 s, int level);
static void inv_lazy_execution(int T, u64 p, int sched = 0)
{
    const int n = 1 << T;
    std::mt19937_64 rng(T);
    std::vector<u64> x(n);
    for (int i = 0; i < n; i++)
        x[i] = (i % 3 == 0) ? 2 * p - 1 : rng() % (2 * p);
    u128 peak = 0;
    bool wrapped = false;
    for (int l = 0; l < T; l++)
    {
        const int gap = 1 << l;
        const u64 addend = p << inv_lazy_shift(T, l, sched);
        for (int blk = 0; blk < n; blk += 2 * gap)
        {
            const u64 w = rng() % p;
            const u64 ws = static_cast<u64>((static_cast<u128>(w) << 64) / p);
            for (int j = blk; j < blk + gap; j++)
            {
                const u64 u = x[j], y = x[j + gap];
                const u128 sum = static_cast<u128>(u) + y;
                const i128 diff = static_cast<i128>(u) - static_cast<i128>(y) + static_cast<i128>(addend);
                wrapped = wrapped || sum >= kWord || diff < 0 || static_cast<u128>(diff) >= kWord;
                peak = sum > peak ? sum : peak;
                peak = static_cast<u128>(diff) > peak ? static_cast<u128>(diff) : peak;
                u64 s = static_cast<u64>(sum);
                if (inv_lazy_mode(T, l, sched) == 2 && sched == 0) // barrett_lazy: x - floor(x * floor(2^64 / p) / 2^64) * p
                    s = s - static_cast<u64>((static_cast<u128>(s) * static_cast<u64>(kWord / p)) >> 64) * p;
                else if (inv_lazy_mode(T, l, sched) == 2) // dense schedule: the single-precision quotient estimate (section 6)
                {
                    const float c = static_cast<float>(4294967296.0 / static_cast<double>(p) * (1.0 - 0x1p-20));
                    const unsigned q = static_cast<unsigned>(static_cast<float>(static_cast<unsigned>(s >> 32)) * c);
                    s = s - static_cast<u64>(q) * p;
                    wrapped = wrapped || sum >= static_cast<u128>(kInvDenseSumMult) * p;
                }
                const u64 d = static_cast<u64>(diff);
                // MODE 1 layers: the level-2 quotient (product below 4p); reducing layers: the exact one (below 2p)
                const u64 q = quotient_model(d, ws, inv_lazy_mode(T, l, sched) == 1 ? 2 : 0);
                x[j] = s;
                x[j + gap] = d * w - q * p;
                wrapped = wrapped || static_cast<u128>(d) * w - static_cast<u128>(q) * p >=
                                         static_cast<u128>(inv_lazy_mode(T, l, sched) == 1 ? kInvLazyProductMult : 2) * p;
            }
        }
    }
    CHECK(!wrapped, "inverse lazy execution T=%d p=%llu sched=%d wrapped", T, p, sched);
    CHECK(peak <= inv_lazy_peak(T, p, sched), "inverse lazy execution T=%d sched=%d above the recurrence", T, sched);
    bool below_2p = true;
    for (int i = 0; i < n; i++)
        below_2p = below_2p && x[i] < 2 * p;
    CHECK(below_2p, "inverse lazy execution T=%d: outputs must be below 2p", T);
}

// ---- the approximate Shoup quotients of devmath.hpp (mulhi_apx: level 1, mulhi_apx2: level 2) as 32-bit limb arithmetic
static u64 quotient_model(u64 y, u64 s, int level)
{
    const u64 y0 = y & 0xFFFFFFFFu, y1 = y >> 32, s0 = s & 0xFFFFFFFFu, s1 = s >> 32;
    if (level == 0)
        return static_cast<u64>((static_cast<u128>(y) * s) >> 64);
    if (level == 1) // exact middle sum of the two cross products (carry kept), hi32(y0 s0) dropped
        return static_cast<u64>(static_cast<u128>(y1) * s1 + ((static_cast<u128>(y1) * s0 + static_cast<u128>(y0) * s1) >> 32));
    return y1 * s1 + ((y0 * s1) >> 32) + ((y1 * s0) >> 32); // level 2: no carry anywhere (the u64 sum cannot wrap: q <= y s / 2^64)
}
static void quotient_shortfall_check()
{
    std::mt19937_64 rng(4);
    for (int level = 0; level <= 2; level++)
    {
        int worst = 0;
        for (int bits = 30; bits <= kMaxPrimeBits; bits++)
            for (int it = 0; it < 4000; it++)
            {
                const u64 p = (it % 7 == 0) ? max_prime_of_bits(bits) : ((u64(1) << (bits - 1)) | (rng() >> (65 - bits)) | 1);
                u64 w = rng() % p, y = rng();
                switch (it % 5) // adversarial limbs: all-ones halves make every dropped piece as large as it gets
                {
                case 0: w = p - 1; y = ~u64(0); break;
                case 1: y |= 0xFFFFFFFFu; break;
                case 2: y = (y << 32) | 0xFFFFFFFFu; break;
                case 3: y >>= (it % 31); break;
                default: break;
                }
                const u64 sh = static_cast<u64>((static_cast<u128>(w) << 64) / p);
                const u64 q = quotient_model(y, sh, level);
                const u128 prod = static_cast<u128>(y) * w;
                const u128 qt = prod / p;
                CHECK(q <= qt, "quotient level %d overestimates", level);
                const int sf = static_cast<int>(qt - q);
                worst = sf > worst ? sf : worst;
                CHECK(sf <= quotient_shortfall(level), "quotient level %d: shortfall %d above the documented %d", level, sf,
                      quotient_shortfall(level));
                const u128 r = prod - static_cast<u128>(q) * p;
                CHECK(r < static_cast<u128>(quotient_shortfall(level) + 1) * p, "quotient level %d: product not below (shortfall + 1) p", level);
                // what the kernel computes mod 2^64 is that true value as long as it is below 2^64
                if (r < kWord)
                    CHECK(static_cast<u64>(y * w - q * p) == static_cast<u64>(r), "quotient level %d: wrapped product differs", level);
            }
        CHECK(worst >= quotient_shortfall(level) - 1, "quotient level %d: documented shortfall %d is not nearly reached (worst %d)", level,
              quotient_shortfall(level), worst);
    }
}

// ---- forward integer schedule with the approximate quotient executed on 64-bit words, true values shadowed in 128 bits:
// inputs below 2p, log n layers of u' = u + v, y' = u - v + g p with v = y w - q p, no reduction anywhere (kNttAnyRep)
static void fwd_apx_execution(int logn, u64 p)
{
    const int n = 1 << logn;
    std::mt19937_64 rng(logn * 7 + 1);
    std::vector<u64> x(n), shadow(n);
    for (int i = 0; i < n; i++)
    {
        x[i] = (i % 3 == 0) ? 2 * p - 1 : rng() % (2 * p);
        shadow[i] = x[i] % p;
    }
    const u64 g = static_cast<u64>(kFwdApxGrowth) * p;
    u128 peak = 0;
    bool wrapped = false;
    for (int l = logn - 1; l >= 0; l--)
    {
        const int gap = 1 << l;
        for (int blk = 0; blk < n; blk += 2 * gap)
        {
            const u64 w = (blk / (2 * gap)) % 5 == 0 ? p - 1 : rng() % p;
            const u64 ws = static_cast<u64>((static_cast<u128>(w) << 64) / p);
            for (int j = blk; j < blk + gap; j++)
            {
                const u64 u = x[j], y = x[j + gap];
                const u64 q = quotient_model(y, ws, kFwdApxLevel);
                const u128 v = static_cast<u128>(y) * w - static_cast<u128>(q) * p;
                const u128 X = static_cast<u128>(u) + v;
                const i128 Y = static_cast<i128>(u) - static_cast<i128>(v) + g;
                wrapped = wrapped || v >= g || X >= kWord || Y < 0 || static_cast<u128>(Y) >= kWord;
                peak = X > peak ? X : peak;
                peak = static_cast<u128>(Y) > peak ? static_cast<u128>(Y) : peak;
                x[j] = u + (y * w - q * p);               // what the kernel's multiply-accumulate chain leaves
                x[j + gap] = (u << 1) + g - x[j];         // y' = 2u + g p - X
                const u64 sv = static_cast<u64>(static_cast<u128>(shadow[j + gap]) * w % p), su = shadow[j];
                shadow[j] = (su + sv) % p;
                shadow[j + gap] = (su + p - sv) % p;
            }
        }
    }
    CHECK(!wrapped, "forward approximate execution logn=%d p=%llu wrapped", logn, p);
    CHECK(peak <= fwd_int_peak(logn, p, 2, true, true), "forward approximate execution logn=%d above the recurrence", logn);
    int bad = 0;
    for (int i = 0; i < n; i++)
        bad += x[i] % p != shadow[i] || x[i] >= static_cast<u128>(fwd_apx_output_mult(logn, true)) * p;
    CHECK(bad == 0, "forward approximate execution logn=%d: %d words leave their residue class or the documented range", logn, bad);
}

// ---- dense lazy forward schedule (ntt_bounds.hpp section 2b; STRICT mode's 56-60-bit rows) executed on 64-bit words with
// the true values shadowed in 128 bits: inputs below 4p, the reference's butterfly with the EXACT Shoup quotient and no
// reduction of its first operand, every word brought below 2p by the single-precision quotient estimate before rounds 2 and
// 3 and in the store -- the layers in the order and grouping of ntt_fwd_half_kernel (top layer, three rounds of four, final)
static void fwd_dense_execution(int logn, u64 p)
{
    const int n = 1 << logn;
    std::mt19937_64 rng(logn * 11 + 3);
    std::vector<u64> x(n), shadow(n);
    for (int i = 0; i < n; i++)
    {
        x[i] = (i % 3 == 0) ? kFwdDenseInMult * p - 1 : rng() % (kFwdDenseInMult * p);
        shadow[i] = x[i] % p;
    }
    const float c = static_cast<float>(4294967296.0 / static_cast<double>(p) * (1.0 - 0x1p-20));
    u128 peak = 0;
    bool wrapped = false, bad_reduce = false;
    const auto reduce_all = [&] {
        for (int i = 0; i < n; i++)
        {
            const unsigned q = static_cast<unsigned>(static_cast<float>(static_cast<unsigned>(x[i] >> 32)) * c);
            const u64 r = x[i] - static_cast<u64>(q) * p;
            bad_reduce = bad_reduce || r >= 2 * p || r % p != x[i] % p;
            x[i] = r;
        }
    };
    int layer = 0; // 0 = the top layer (gap n/2); the half-row kernel's rounds are layers 1-4, 5-8, 9-12, the rest is the final round
    for (int l = logn - 1; l >= 0; l--, layer++)
    {
        const int round = layer == 0 ? 0 : (layer <= 12 ? (layer - 1) / 4 + 1 : 4);
        if (layer >= 1 && layer <= 12 && (layer - 1) % 4 == 0 && fwd_dense_reduce_before_round(round))
            reduce_all();
        const int gap = 1 << l;
        for (int blk = 0; blk < n; blk += 2 * gap)
        {
            const u64 w = (blk / (2 * gap)) % 5 == 0 ? p - 1 : rng() % p;
            const u64 ws = static_cast<u64>((static_cast<u128>(w) << 64) / p);
            for (int j = blk; j < blk + gap; j++)
            {
                const u64 u = x[j], y = x[j + gap];
                const u64 q = quotient_model(y, ws, 0);
                const u128 v = static_cast<u128>(y) * w - static_cast<u128>(q) * p;
                const u128 X = static_cast<u128>(u) + v;
                const i128 Y = static_cast<i128>(u) - static_cast<i128>(v) + 2 * static_cast<i128>(p);
                wrapped = wrapped || v >= 2 * static_cast<u128>(p) || X >= kWord || Y < 0 || static_cast<u128>(Y) >= kWord;
                peak = X > peak ? X : peak;
                peak = static_cast<u128>(Y) > peak ? static_cast<u128>(Y) : peak;
                x[j] = u + (y * w - q * p);
                x[j + gap] = (u << 1) + 2 * p - x[j];
                const u64 sv = static_cast<u64>(static_cast<u128>(shadow[j + gap]) * w % p), su = shadow[j];
                shadow[j] = (su + sv) % p;
                shadow[j + gap] = (su + p - sv) % p;
            }
        }
    }
    reduce_all(); // the store
    CHECK(!wrapped && !bad_reduce, "dense forward execution logn=%d p=%llu wrapped or a reduction left [0, 2p)", logn, p);
    CHECK(peak < static_cast<u128>(fwd_dense_peak_mult(logn)) * p, "dense forward execution logn=%d above its bound", logn);
    int bad = 0;
    for (int i = 0; i < n; i++)
        bad += x[i] % p != shadow[i] || x[i] >= 2 * p;
    CHECK(bad == 0, "dense forward execution logn=%d: %d words leave their residue class or [0, 2p)", logn, bad);
}

// ---- devmath.hpp reduce_small_quot: the same IEEE single-precision operations (u32 -> float round to nearest, one
// multiplication, truncation) on words that sit on and next to multiples of p
static void small_quot_model()
{
    std::mt19937_64 rng(6);
    int n = 0;
    for (int bits = 45; bits <= 58; bits++)
        for (int it = 0; it < 3000; it++)
        {
            const u64 p = (it % 5 == 0) ? max_prime_of_bits(bits) : ((u64(1) << (bits - 1)) | (rng() >> (65 - bits)) | 1);
            for (int mult : { 2, 6, 50, 62, 66, 128 })
            {
                if (!small_quot_admits(p, mult))
                    continue;
                const float c = static_cast<float>(4294967296.0 / static_cast<double>(p) * (1.0 - 0x1p-20));
                const u64 k = rng() % mult; // x = k p + d with d at the edges and in the middle
                const u64 ds[6] = { 0, 1, p - 1, p / 2, rng() % p, (rng() % p) | 0xFFFFFFFFu };
                for (u64 d : ds)
                {
                    if (d >= p)
                        continue;
                    const u64 x = k * p + d;
                    const unsigned q = static_cast<unsigned>(static_cast<float>(static_cast<unsigned>(x >> 32)) * c);
                    CHECK(q <= k && q + 1 >= k, "small quotient: p=%llu x=%llu estimate %u, true %llu", p, x, q, k);
                    const u64 r = x - static_cast<u64>(q) * p;
                    CHECK(r < 2 * p && r % p == d, "small quotient: reduced word out of [0, 2p) or off its class");
                    n++;
                }
            }
        }
    CHECK(n > 100000, "small quotient model ran %d cases", n);
    CHECK(!small_quot_admits((u64(1) << 45) - 1, 66) && small_quot_admits(u64(1) << 45, 66) && !small_quot_admits(max_prime_of_bits(58), 129),
          "small quotient admission edges");
}

// ---- devmath.hpp DotAcc<NTERMS> executed on the largest operands: every accumulator tracked in 128 bits
static void dotacc_execution()
{
    for (int n = 1; n <= 64; n++)
    {
        CHECK(dotacc_ok(n, kDotAccOperandBits), "DotAcc<%d> on 61-bit operands should be admitted", n);
        const int nm = (2 * n + 7) / 8;
        const u64 t = (u64(1) << 61) - 1, c = (u64(1) << 61) - 1; // all operands at the top of the admitted range
        u128 l0 = 0, l1 = 0, h = 0, m[16] = {};
        u128 exact = 0;
        for (int i = 0; i < n; i++)
        {
            const u64 t0 = t & 0xFFFFFFFFu, t1 = t >> 32, t00 = t0 & 0xFFFFu, t01 = t0 >> 16, c0 = c & 0xFFFFFFFFu, c1 = c >> 32;
            l0 += static_cast<u128>(t00) * c0;
            l1 += static_cast<u128>(t01) * c0;
            m[(2 * i) % nm] += static_cast<u128>(t0) * c1;
            m[(2 * i + 1) % nm] += static_cast<u128>(t1) * c0;
            h += static_cast<u128>(t1) * c1;
            exact += static_cast<u128>(t) * c;
        }
        bool fits = l0 < kWord && l1 < kWord && h < kWord;
        u128 sum = (h << 64) + l0 + (l1 << 16);
        for (int i = 0; i < nm; i++)
        {
            fits = fits && m[i] < kWord;
            sum += m[i] << 32;
        }
        CHECK(fits, "DotAcc<%d>: an accumulator passes 2^64 on 61-bit operands", n);
        CHECK(sum == exact, "DotAcc<%d>: assembled sum differs from the exact one", n);
    }
    // tight: one more operand bit and the middle accumulators (eight products each at NTERMS = 8) no longer fit
    CHECK(!dotacc_ok(8, kDotAccOperandBits + 1), "DotAcc predicate not tight at 62 bits");
}

int main()
{
    std::fesetround(FE_TONEAREST);
    dotacc_execution();
    enumerate_predicates();
    quotient_shortfall_check();
    small_quot_model();
    for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
        fwd_apx_execution(logn, max_prime_of_bits(fwd_lazy_prime_bits(logn)));
    for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
    {
        fwd_dense_execution(logn, max_prime_of_bits(60));
        fwd_dense_execution(logn, (u64(1) << 59) + 12345);
        fwd_dense_execution(logn, (u64(1) << 45) + 7);
        CHECK(fwd_dense_admits(max_prime_of_bits(60), logn) && fwd_dense_admits(u64(1) << 45, logn) &&
                  !fwd_dense_admits((u64(1) << 45) - 1, logn) && !fwd_dense_admits((u64(1) << 60) + 1, logn),
              "dense forward admission edges logn=%d", logn);
    }
    regression_whole_row();
    fp_product_model();
    for (int logn = kMinHalfLogn; logn <= kMaxHalfLogn; logn++)
        for (int pattern = 0; pattern < 4; pattern++)
            fp_forward_execution(logn, max_ntt_prime_of_bits(kFpPrimeBits, logn), pattern);
    for (int T = kMinHalfLogn - 1; T <= kMaxHalfLogn; T++)
    {
        inv_lazy_execution(T, max_prime_of_bits(inv_lazy_prime_bits(T)));
        inv_lazy_execution(T, max_prime_of_bits(inv_lazy_prime_bits(T, 1)), 1); // dense schedule at the largest 60-bit value
        inv_lazy_execution(T, (u64(1) << 59) + 12345, 1);
    }
    std::printf(failures ? "bounds_check: %d FAILURES\n" : "bounds_check: OK\n", failures);
    return failures ? 1 : 0;
}
