// hostmath.cpp -- see hostmath.hpp
#include "hostmath.hpp"

#include <cmath>

#include <algorithm>

namespace sealhip
{
    HostModulus::HostModulus(u64 v) : value(v)
    {
        if (v < 2 || (v >> 61) != 0)
            throw std::invalid_argument("modulus must be in [2, 2^61)"); // modulus.cpp:77-80
        for (u64 x = v; x; x >>= 1)
            bits++;
        // floor(2^128 / v) = floor((2^128 - 1) / v) unless v divides 2^128 (v a power of two)
        u128 all = ~static_cast<u128>(0);
        u128 q = all / v;
        if (all % v == static_cast<u128>(v - 1))
            q += 1;
        cr0 = static_cast<u64>(q);
        cr1 = static_cast<u64>(q >> 64);
    }

    u64 powmod(u64 a, u64 e, u64 p)
    {
        u64 r = 1 % p;
        a %= p;
        while (e)
        {
            if (e & 1)
                r = mulmod(r, a, p);
            a = mulmod(a, a, p);
            e >>= 1;
        }
        return r;
    }

    bool invmod(u64 a, u64 p, u64 &out)
    {
        a %= p;
        if (a == 0)
            return false;
        // extended Euclid on (a, p) with signed 128-bit cofactors
        __int128 t0 = 0, t1 = 1;
        u64 r0 = p, r1 = a;
        while (r1 != 0)
        {
            u64 qq = r0 / r1;
            u64 r2 = r0 - qq * r1;
            __int128 t2 = t0 - static_cast<__int128>(qq) * t1;
            r0 = r1;
            r1 = r2;
            t0 = t1;
            t1 = t2;
        }
        if (r0 != 1)
            return false;
        __int128 res = t0 % static_cast<__int128>(p);
        if (res < 0)
            res += p;
        out = static_cast<u64>(res);
        return true;
    }

    bool is_prime_u64(u64 n)
    {
        if (n < 2)
            return false;
        static const u64 witnesses[] = { 2, 3, 5, 7, 11, 13, 17, 19, 23, 29, 31, 37 };
        for (u64 w : witnesses)
        {
            if (n == w)
                return true;
            if (n % w == 0)
                return false;
        }
        u64 d = n - 1;
        int s = 0;
        while ((d & 1) == 0)
        {
            d >>= 1;
            s++;
        }
        for (u64 w : witnesses)
        {
            u64 x = powmod(w, d, n);
            if (x == 1 || x == n - 1)
                continue;
            bool witness_composite = true;
            for (int i = 1; i < s; i++)
            {
                x = mulmod(x, x, n);
                if (x == n - 1)
                {
                    witness_composite = false;
                    break;
                }
            }
            if (witness_composite)
                return false;
        }
        return true;
    }

    std::vector<u64> get_primes(std::size_t ntt_size, int bit_size, std::size_t count)
    {
        std::vector<u64> out;
        u64 step = 2 * static_cast<u64>(ntt_size);
        u64 top = u64(1) << bit_size;
        if (top < step)
            throw std::logic_error("failed to find enough qualifying primes");
        u64 cand = top - step + 1;
        u64 floor_value = u64(1) << (bit_size - 1);
        while (out.size() < count && cand > floor_value)
        {
            if (is_prime_u64(cand))
                out.push_back(cand);
            cand -= step;
        }
        if (out.size() < count)
            throw std::logic_error("failed to find enough qualifying primes");
        return out;
    }

    bool minimal_primitive_root(u64 degree, u64 p, u64 &root)
    {
        if ((p - 1) % degree != 0)
            return false;
        u64 cofactor = (p - 1) / degree;
        u64 g = 0;
        for (u64 c = 2; c < p; c++)
        {
            u64 r = powmod(c, cofactor, p);
            if (r != 0 && powmod(r, degree >> 1, p) == p - 1)
            {
                g = r;
                break;
            }
            if (c > 200000)
                return false;
        }
        if (!g)
            return false;
        // all primitive roots are the odd powers of g; keep the smallest
        u64 g2 = mulmod(g, g, p), cur = g, best = g;
        for (u64 i = 0; i < degree / 2 + 1; i++)
        {
            best = std::min(best, cur);
            cur = mulmod(cur, g2, p);
        }
        root = best;
        return true;
    }

    std::uint32_t reverse_bits(std::uint32_t x, int bit_count)
    {
        std::uint32_t r = 0;
        for (int i = 0; i < bit_count; i++)
            r |= ((x >> i) & 1u) << (bit_count - 1 - i);
        return r;
    }

    void complex_root(std::size_t m, std::size_t index, double &re, double &im)
    {
        static const double PI_ = 3.1415926535897932384626433832795028842; // croots.h:27
        index &= m - 1;
        double a, b;
        if (index <= m / 8)
        {
            const double th = 2 * PI_ * static_cast<double>(index) / static_cast<double>(m); // std::polar(1.0, th)
            // one libm entry point for both parts: glibc's sincos and sin/cos differ in the last bit for a few angles (1 of
            // 1025 at m = 8192), and an optimising compiler picks either for std::polar; the oracle calls sincos too
            ::sincos(th, &im, &re);
        }
        else if (index <= m / 4)
        {
            complex_root(m, m / 4 - index, a, b);
            re = b;
            im = a;
        }
        else if (index <= m / 2)
        {
            complex_root(m, m / 2 - index, a, b);
            re = -a;
            im = b;
        }
        else if (index <= 3 * m / 4)
        {
            complex_root(m, index - m / 2, a, b);
            re = -a;
            im = -b;
        }
        else
        {
            complex_root(m, m - index, a, b);
            re = a;
            im = -b;
        }
    }

    void HostNttTables::build(int logn_, u64 p_)
    {
        logn = logn_;
        p = p_;
        const std::size_t n = std::size_t(1) << logn;
        if (!minimal_primitive_root(2 * static_cast<u64>(n), p, psi))
            throw std::invalid_argument("invalid modulus: no primitive 2N-th root");
        u64 psi_inv;
        if (!invmod(psi, p, psi_inv) || !invmod(static_cast<u64>(n) % p, p, inv_n))
            throw std::invalid_argument("invalid modulus");
        inv_n_shoup = shoup(inv_n, p);
        rdp = shoup(1, p);
        fwd.assign(2 * n, 0);
        inv.assign(2 * n, 0);
        u64 pw = 1, pwi = 1;
        for (std::size_t e = 0; e < n; e++)
        {
            std::size_t idx = reverse_bits(static_cast<std::uint32_t>(e), logn);
            fwd[2 * idx] = pw;
            fwd[2 * idx + 1] = shoup(pw, p);
            inv[2 * idx] = pwi;
            inv[2 * idx + 1] = shoup(pwi, p);
            pw = mulmod(pw, psi, p);
            pwi = mulmod(pwi, psi_inv, p);
        }
        // the single twiddle of the top inverse layer, with n^{-1} folded in
        u64 w_top = n > 1 ? inv[2 * 1] : 1;
        inv_n_w = mulmod(w_top, inv_n, p);
        inv_n_w_shoup = shoup(inv_n_w, p);
    }

    std::vector<u64> HostNttTables::reference_table(int kind) const
    {
        const std::size_t n = std::size_t(1) << logn;
        std::vector<u64> out(n);
        if (kind == 0 || kind == 1)
        {
            for (std::size_t i = 0; i < n; i++)
                out[i] = fwd[2 * i + kind];
            return out;
        }
        // ntt.cpp:84-98: positions 1..N-1 are the concatenation for m = N/2, N/4, ..., 1 of [m, 2m),
        // last entry multiplied by n^{-1}
        std::vector<u64> w(n);
        w[0] = inv[0];
        std::size_t pos = 1;
        for (std::size_t m = n >> 1; m > 0; m >>= 1)
            for (std::size_t i = 0; i < m; i++)
                w[pos++] = inv[2 * (m + i)];
        if (n > 1)
            w[n - 1] = mulmod(w[n - 1], inv_n, p);
        if (kind == 2)
            return w;
        for (std::size_t i = 0; i < n; i++)
            out[i] = shoup(w[i], p);
        return out;
    }

    static u64 product_mod(const std::vector<u64> &vals, std::size_t count, u64 p)
    {
        u64 r = 1 % p;
        for (std::size_t i = 0; i < count; i++)
            r = mulmod(r, vals[i] % p, p);
        return r;
    }

    void HostBaseConv::build(const std::vector<u64> &ib, const std::vector<u64> &ob)
    {
        ibase = ib;
        obase = ob;
        const std::size_t is = ib.size(), os = ob.size();
        inv_punct.assign(is, 1);
        matrix.assign(is * os, 0);
        for (std::size_t i = 0; i < is; i++)
        {
            if (is == 1)
                break; // rns.cpp:281-286
            u64 prod = 1 % ib[i];
            for (std::size_t l = 0; l < is; l++)
                if (l != i)
                    prod = mulmod(prod, ib[l] % ib[i], ib[i]);
            if (!invmod(prod, ib[i], inv_punct[i]))
                throw std::logic_error("invalid rns bases");
        }
        for (std::size_t j = 0; j < os; j++)
            for (std::size_t i = 0; i < is; i++)
            {
                u64 prod = 1 % ob[j];
                for (std::size_t l = 0; l < is; l++)
                    if (l != i)
                        prod = mulmod(prod, ib[l] % ob[j], ob[j]);
                matrix[j * is + i] = prod;
            }
    }

    static int product_bit_length(const std::vector<u64> &vals)
    {
        std::vector<u64> acc(1, 1);
        for (u64 v : vals)
        {
            u64 carry = 0;
            for (auto &word : acc)
            {
                u128 z = static_cast<u128>(word) * v + carry;
                word = static_cast<u64>(z);
                carry = static_cast<u64>(z >> 64);
            }
            if (carry)
                acc.push_back(carry);
        }
        int bits = 0;
        for (u64 x = acc.back(); x; x >>= 1)
            bits++;
        return bits + 64 * static_cast<int>(acc.size() - 1);
    }

    std::size_t HostRnsTool::base_B_size(const std::vector<u64> &q, u64 t)
    {
        int t_bits = 0;
        for (u64 x = t; x; x >>= 1)
            t_bits++;
        std::size_t b = q.size();
        // rns.cpp:568-573, SEAL_INTERNAL_MOD_BIT_COUNT = 61
        if (32 + t_bits + product_bit_length(q) >= 61 * static_cast<int>(q.size()) + 61)
            b++;
        return b;
    }

    void HostRnsTool::build(std::size_t n_, const std::vector<u64> &q_, u64 t_, const std::vector<u64> &aux)
    {
        n = n_;
        q = q_;
        t = t_;
        B_size = base_B_size(q, t);
        if (aux.size() < B_size + 2)
            throw std::logic_error("not enough auxiliary primes");
        m_sk = aux[0]; // rns.cpp:589-592
        gamma = aux[1];
        std::vector<u64> B(aux.begin() + 2, aux.begin() + 2 + static_cast<std::ptrdiff_t>(B_size));
        Bsk = B;
        Bsk.push_back(m_sk);
        q_to_Bsk.build(q, Bsk);
        q_to_m_tilde.build(q, std::vector<u64>{ m_tilde });
        B_to_q.build(B, q);
        B_to_m_sk.build(B, std::vector<u64>{ m_sk });
        const std::size_t k = q.size(), nb = Bsk.size();
        prod_B_mod_q.resize(k);
        for (std::size_t i = 0; i < k; i++)
            prod_B_mod_q[i] = product_mod(B, B.size(), q[i]);
        inv_prod_q_mod_Bsk.resize(nb);
        prod_q_mod_Bsk.resize(nb);
        inv_m_tilde_mod_Bsk.resize(nb);
        for (std::size_t i = 0; i < nb; i++)
        {
            prod_q_mod_Bsk[i] = product_mod(q, k, Bsk[i]);
            if (!invmod(prod_q_mod_Bsk[i], Bsk[i], inv_prod_q_mod_Bsk[i]) ||
                !invmod(m_tilde % Bsk[i], Bsk[i], inv_m_tilde_mod_Bsk[i]))
                throw std::logic_error("invalid rns bases");
        }
        if (!invmod(product_mod(B, B.size(), m_sk), m_sk, inv_prod_B_mod_m_sk) ||
            !invmod(product_mod(q, k, m_tilde), m_tilde, inv_prod_q_mod_m_tilde))
            throw std::logic_error("invalid rns bases");
        inv_q_last_mod_q.assign(k > 0 ? k - 1 : 0, 0);
        for (std::size_t i = 0; i + 1 < k; i++)
            if (!invmod(q[k - 1], q[i], inv_q_last_mod_q[i]))
                throw std::logic_error("invalid rns bases");
    }
} // namespace sealhip
