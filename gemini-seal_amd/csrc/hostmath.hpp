// hostmath.hpp -- host-side number theory and table/constant generation for the sealhip engine.
//
// The GPU box never sees a SEALContext, so everything seal::SEALContext would have precomputed for
// the hot path is regenerated here from plain parameters: NTT tables (native/src/seal/util/ntt.cpp:37-99),
// prime selection (util/numth.cpp:277-323), the minimal primitive root (util/numth.cpp:398-424),
// Modulus::const_ratio (modulus.cpp:85-98) and the RNSTool constants (util/rns.cpp:539-729).
// All results are canonical residues, so any exact method yields the reference's values.
#pragma once

#include <cstddef>
#include <cstdint>
#include <stdexcept>
#include <vector>

namespace sealhip
{
    using u64 = unsigned long long; // same width as uint64_t; matches the device-side typedef
    using u128 = unsigned __int128;

    struct HostModulus
    {
        u64 value = 0;
        u64 cr0 = 0, cr1 = 0; // floor(2^128 / value)
        int bits = 0;
        HostModulus() = default;
        explicit HostModulus(u64 v);
    };

    inline u64 mulmod(u64 a, u64 b, u64 p)
    {
        return static_cast<u64>((static_cast<u128>(a) * b) % p);
    }
    u64 powmod(u64 a, u64 e, u64 p);
    bool invmod(u64 a, u64 p, u64 &out);
    inline u64 shoup(u64 x, u64 p)
    {
        return static_cast<u64>((static_cast<u128>(x) << 64) / p);
    }
    bool is_prime_u64(u64 n);
    // largest-first primes = 1 (mod 2*ntt_size) below 2^bit_size (numth.cpp:277-323)
    std::vector<u64> get_primes(std::size_t ntt_size, int bit_size, std::size_t count);
    // smallest of all primitive degree-th roots of unity mod p (numth.cpp:398-424)
    bool minimal_primitive_root(u64 degree, u64 p, u64 &root);
    std::uint32_t reverse_bits(std::uint32_t x, int bit_count);
    // util::ComplexRoots::get_root (util/croots.cpp:17-70): exp(2 pi i index / degree) through the 8-fold symmetry
    void complex_root(std::size_t degree, std::size_t index, double &re, double &im);

    // Twiddle tables of one prime. Device layout (ours, not the reference's): for each direction an
    // array of N pairs {w, floor(w*2^64/p)} indexed by the bit-reversed exponent, i.e. entry i holds
    // psi^{bitrev(i)} (forward) or psi^{-bitrev(i)} (inverse, NOT re-ordered). A butterfly on global
    // bit b whose lower element has index j uses entry (N + j) >> (b + 1) in both directions.
    struct HostNttTables
    {
        int logn = 0;
        u64 p = 0, psi = 0;
        u64 inv_n = 0, inv_n_shoup = 0;     // n^{-1}, shoup(n^{-1})
        u64 inv_n_w = 0, inv_n_w_shoup = 0; // psi^{-bitrev(1)} * n^{-1} for the last inverse layer (ntt.cpp:97)
        u64 rdp = 0;                        // floor(2^64/p) (ntt.cpp:75)
        std::vector<u64> fwd;               // 2N words: {w, w'} pairs
        std::vector<u64> inv;               // 2N words
        void build(int logn_, u64 p_);
        // reference-ordered views for introspection: kind 0..3 as in sealhip_debug_ntt_table
        std::vector<u64> reference_table(int kind) const;
    };

    // approximate base conversion constants (rns.cpp:237-290, 498-523)
    struct HostBaseConv
    {
        std::vector<u64> ibase, obase;
        std::vector<u64> inv_punct; // [isize]
        std::vector<u64> matrix;    // [osize][isize]
        void build(const std::vector<u64> &ib, const std::vector<u64> &ob);
    };

    // everything RNSTool::initialize derives for one level (rns.cpp:539-729)
    struct HostRnsTool
    {
        std::size_t n = 0;
        std::vector<u64> q;   // k primes
        std::vector<u64> Bsk; // B..., m_sk last
        std::size_t B_size = 0;
        u64 m_tilde = u64(1) << 32, m_sk = 0, gamma = 0, t = 0;
        HostBaseConv q_to_Bsk, q_to_m_tilde, B_to_q, B_to_m_sk;
        std::vector<u64> prod_B_mod_q, inv_prod_q_mod_Bsk, prod_q_mod_Bsk, inv_m_tilde_mod_Bsk, inv_q_last_mod_q;
        u64 inv_prod_B_mod_m_sk = 0, inv_prod_q_mod_m_tilde = 0;
        // aux = get_primes(n, 60, count) with count >= |B| + 2
        void build(std::size_t n_, const std::vector<u64> &q_, u64 t_, const std::vector<u64> &aux);
        static std::size_t base_B_size(const std::vector<u64> &q, u64 t);
    };
} // namespace sealhip
