// Compile-and-link check of the C++ host adapter (gemini-seal_amd/host/evaluator.hpp). With a GPU it also runs one
// multiply through the adapter and prints a digest that the Python test compares with the oracle.
#include <cstdio>
#include <cstdlib>

#include "../gemini-seal_amd/host/evaluator.hpp"

using namespace sealhip_host;

static std::uint64_t splitmix(std::uint64_t &s)
{
    std::uint64_t z = (s += 0x9E3779B97F4A7C15ULL);
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ULL;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBULL;
    return z ^ (z >> 31);
}

int main(int argc, char **argv)
{
    // cfg1 of BASELINE.json: BFV N=4096, {36,36,37}
    const std::uint64_t mods[3] = { 68719230977ULL, 68719403009ULL, 137438822401ULL };
    sealhip_params p{ SEALHIP_SCHEME_BFV, 12, 3, 1, mods, 786433, SEALHIP_MODE_PARITY, argc > 1 ? std::atoi(argv[1]) : -1 };
    try
    {
        Context ctx(p);
        if (p.device < 0)
        {
            std::printf("host-only context ok\n");
            return 0;
        }
        const std::size_t n = 4096, k = 2;
        std::uint64_t state = 0xC0FFEE + 1;
        // same fill order as the survey generator (keys first), SURVEY Appendix B.2
        std::vector<std::uint64_t> key(2 * 2 * 3 * n);
        for (std::size_t d = 0; d < 2; d++)
            for (std::size_t l = 0; l < 2; l++)
                for (std::size_t r = 0; r < 3; r++)
                    for (std::size_t c = 0; c < n; c++)
                        key[((d * 2 + l) * 3 + r) * n + c] = splitmix(state) % mods[r];
        HostCiphertext a, b;
        for (HostCiphertext *ct : { &a, &b })
        {
            ct->n_ = n;
            ct->resize_raw(2, k);
            for (std::size_t s = 0; s < 2; s++)
                for (std::size_t r = 0; r < k; r++)
                    for (std::size_t c = 0; c < n; c++)
                        ct->words[(s * k + r) * n + c] = splitmix(state) % mods[r];
        }
        Evaluator<HostCiphertext> ev(ctx);
        KSwitchKeys rk(ctx, key.data(), 2);
        ev.multiply_inplace(a, b);
        ev.relinearize_inplace(a, { &rk });
        ev.mod_switch_to_next_inplace(a);
        std::uint64_t h = 0xcbf29ce484222325ULL;
        for (std::uint64_t w : a.words)
            for (int i = 0; i < 8; i++)
            {
                h ^= (w >> (8 * i)) & 0xff;
                h *= 0x100000001b3ULL;
            }
        std::printf("modswitch digest %016llx\n", (unsigned long long)h);
        // SURVEY 8(f1) methods: algebraic identities through the adapter
        HostCiphertext c = b, d = b;
        ev.add_inplace(c, b);
        ev.sub_inplace(c, b); // (b + b) - b == b
        ev.negate_inplace(d);
        ev.negate_inplace(d); // -(-b) == b
        std::vector<std::uint64_t> one(n, 0);
        one[0] = 1;
        HostCiphertext m = b;
        ev.multiply_plain_inplace(m, one.data(), false); // b * 1 == b
        const bool ok = c.words == b.words && d.words == b.words && m.words == b.words && !ev.is_transparent(b);
        std::printf("f1 identities %s\n", ok ? "ok" : "FAILED");
        if (!ok)
            return 1;
        // A batch of separately allocated ciphertexts (what a vector<Ciphertext> is): multiply + relinearize through the
        // pointer-array entry must give, per item, what the one-at-a-time calls above give. 11 items with a chunk of 4
        // (SEALHIP_HOST_CHUNK=4 from the test) walks the double-buffered staging pipeline through three chunks.
        {
            std::uint64_t st2 = 0xC0FFEE + 1;
            for (std::size_t i = 0; i < key.size(); i++)
                (void)splitmix(st2); // same stream position as after the key fill above
            HostCiphertext a0, b0;
            for (HostCiphertext *ct : { &a0, &b0 })
            {
                ct->n_ = n;
                ct->resize_raw(2, k);
                for (std::size_t s = 0; s < 2; s++)
                    for (std::size_t r = 0; r < k; r++)
                        for (std::size_t c2 = 0; c2 < n; c2++)
                            ct->words[(s * k + r) * n + c2] = splitmix(st2) % mods[r];
            }
            HostCiphertext one_by_one = a0;
            ev.multiply_inplace(one_by_one, b0);
            ev.relinearize_inplace(one_by_one, { &rk });
            const std::size_t count = 11;
            std::vector<HostCiphertext> xs(count, a0), ys(count, b0);
            // make the items differ (and scatter the allocations): item i gets its first word bumped by i
            for (std::size_t i = 0; i < count; i++)
                xs[i].words[0] = (xs[i].words[0] + i) % mods[0];
            std::vector<HostCiphertext *> px;
            std::vector<const HostCiphertext *> py;
            for (std::size_t i = 0; i < count; i++)
            {
                px.push_back(&xs[i]);
                py.push_back(&ys[i]);
            }
            const std::vector<const KSwitchKeys *> rks{ &rk };
            ev.multiply_inplace(px, py, &rks);
            bool batch_ok = xs[0].words == one_by_one.words && xs[0].size() == 2;
            for (std::size_t i = 1; i < count && batch_ok; i++)
            {
                HostCiphertext ref = a0;
                ref.words[0] = (ref.words[0] + i) % mods[0];
                ev.multiply_inplace(ref, b0);
                ev.relinearize_inplace(ref, { &rk });
                batch_ok = xs[i].words == ref.words;
            }
            std::printf("host batch %s\n", batch_ok ? "ok" : "FAILED");
            if (!batch_ok)
                return 1;
            // the batch rotated in place, once on pageable buffers and once with every buffer pinned in place
            // (register_pool_block): same words
            {
                std::vector<HostCiphertext> plain(xs), pinned(xs);
                std::vector<HostCiphertext *> pp, pq;
                for (std::size_t i = 0; i < count; i++)
                {
                    pp.push_back(&plain[i]);
                    pq.push_back(&pinned[i]);
                }
                std::uint32_t g1 = 0;
                throw_on(sealhip_galois_elt_from_step(ctx.get(), 1, &g1));
                const std::map<std::uint32_t, const KSwitchKeys *> gk1{ { g1, &rk } };
                ev.rotate_vector_inplace(pp, 1, gk1); // (in place, no reallocation: the pinned pointers stay the objects' buffers)
                for (auto &c : pinned)
                    ev.register_pool_block(c.data(), c.words.size() * sizeof(std::uint64_t));
                ev.rotate_vector_inplace(pq, 1, gk1);
                bool reg_ok = true, refused = false;
                try { ev.register_pool_block(pinned[0].data(), 64); } catch (const std::invalid_argument &) { refused = true; }
                for (auto &c : pinned)
                    ev.unregister_pool_block(c.data());
                for (std::size_t i = 0; i < count; i++)
                    reg_ok = reg_ok && plain[i].words == pinned[i].words && plain[i].words != xs[i].words;
                std::printf("registered batch %s\n", reg_ok && refused ? "ok" : "FAILED");
                if (!reg_ok || !refused)
                    return 1;
            }
            // multiply_many / exponentiate_inplace through the adapter: x^2 == relinearize(x * x), and a three-operand product
            // follows the reference's queue order [a0*b0, a0] -> (a0*b0)*a0
            HostCiphertext sq = a0, want_sq = a0;
            ev.exponentiate_inplace(sq, 2, rks);
            ev.multiply_inplace(want_sq, a0);
            ev.relinearize_inplace(want_sq, { &rk });
            HostCiphertext many, want_many = one_by_one; // one_by_one = relin(a0 * b0)
            ev.multiply_many({ a0, b0, a0 }, rks, many);
            ev.multiply_inplace(want_many, a0);
            ev.relinearize_inplace(want_many, { &rk });
            bool threw = false;
            try
            {
                ev.exponentiate_inplace(sq, 0, rks);
            }
            catch (const std::invalid_argument &)
            {
                threw = true;
            }
            const bool many_ok = sq.words == want_sq.words && many.words == want_many.words && threw;
            std::printf("multiply_many %s\n", many_ok ? "ok" : "FAILED");
            if (!many_ok)
                return 1;
        }
        // Round 4: the rest of f1's names. BFV rotate_rows / rotate_columns on `b` (coefficient form) with
        // the key buffer above standing in for the Galois keys of both elements; digests compared by the Python test with
        // the oracle's apply_galois on the same words. Destination-taking variants, add_many, mod_switch_to.
        {
            const auto digest = [](const HostCiphertext &ct) {
                std::uint64_t hh = 0xcbf29ce484222325ULL;
                for (std::uint64_t w : ct.words)
                    for (int i = 0; i < 8; i++)
                    {
                        hh ^= (w >> (8 * i)) & 0xff;
                        hh *= 0x100000001b3ULL;
                    }
                return (unsigned long long)hh;
            };
            std::uint32_t e1 = 0, ec = 0;
            throw_on(sealhip_galois_elt_from_step(ctx.get(), 1, &e1));
            throw_on(sealhip_galois_elt_from_step(ctx.get(), 0, &ec));
            const std::map<std::uint32_t, const KSwitchKeys *> gks{ { e1, &rk }, { ec, &rk } };
            HostCiphertext rr, rc, rr2 = b;
            ev.rotate_rows(b, 1, gks, rr);
            ev.rotate_columns(b, gks, rc);
            ev.rotate_rows_inplace(rr2, 1, gks);
            std::printf("rotate_rows digest %016llx\n", digest(rr));
            std::printf("rotate_columns digest %016llx\n", digest(rc));
            bool names_ok = rr2.words == rr.words && rr.words != b.words && rc.words != b.words;
            // wrong scheme -> std::logic_error (evaluator.h:1205-1208, :1272-1275)
            int logic = 0;
            try { HostCiphertext t = b; ev.rotate_vector_inplace(t, 1, gks); } catch (const std::logic_error &) { logic++; }
            try { HostCiphertext t = b; ev.complex_conjugate_inplace(t, gks); } catch (const std::logic_error &) { logic++; }
            // a missing key -> std::invalid_argument (evaluator.cpp:1871-1874)
            try { HostCiphertext t = b; ev.rotate_columns_inplace(t, { { e1, &rk } }); } catch (const std::invalid_argument &) { logic++; }
            names_ok = names_ok && logic == 3;
            // destination variants == in-place forms; aliasing of the second operand as the reference handles it
            HostCiphertext s1, s2 = b, d1, n1, m1, m2 = a;   // a is the level-1 result of the chain above: use fresh copies
            HostCiphertext x = b, y = b;
            y.words[1] = (y.words[1] + 5) % mods[0];
            ev.add(x, y, s1);
            HostCiphertext s3 = y;
            ev.add(x, s3, s3); // destination aliases encrypted2
            HostCiphertext want = x;
            ev.add_inplace(want, y);
            names_ok = names_ok && s1.words == want.words && s3.words == want.words;
            ev.sub(x, y, d1);
            HostCiphertext d3 = y;
            ev.sub(x, d3, d3);
            want = x;
            ev.sub_inplace(want, y);
            names_ok = names_ok && d1.words == want.words && d3.words == want.words;
            ev.negate(x, n1);
            want = x;
            ev.negate_inplace(want);
            names_ok = names_ok && n1.words == want.words;
            ev.multiply(x, y, m1);
            want = x;
            ev.multiply_inplace(want, y);
            names_ok = names_ok && m1.words == want.words && m1.size() == 3;
            HostCiphertext rl, ms, ms2, sqd;
            ev.relinearize(m1, { &rk }, rl);
            ev.mod_switch_to_next(rl, ms);
            ev.mod_switch_to(rl, 1, ms2); // level k = 2 -> 1: one step
            ev.square(x, sqd);
            HostCiphertext sq_want = x;
            ev.square_inplace(sq_want);
            names_ok = names_ok && rl.size() == 2 && ms.words == ms2.words && ms.coeff_modulus_size() == 1 && sqd.words == sq_want.words;
            bool higher = false;
            try { ev.mod_switch_to_inplace(ms, 2); } catch (const std::invalid_argument &) { higher = true; } // :1051-1054
            HostCiphertext am, am_want = x;
            ev.add_many({ x, y, x }, am);
            ev.add_inplace(am_want, y);
            ev.add_inplace(am_want, x);
            names_ok = names_ok && higher && am.words == am_want.words;
            std::printf("f1 names %s\n", names_ok ? "ok" : "FAILED");
            if (!names_ok)
                return 1;
        }
        // CKKS: complex_conjugate / rotate_vector / rescale_to through the adapter, same moduli, NTT-form words
        {
            sealhip_params pc{ SEALHIP_SCHEME_CKKS, 12, 3, 1, mods, 0, SEALHIP_MODE_PARITY, p.device };
            Context cctx(pc);
            Evaluator<HostCiphertext> cev(cctx);
            KSwitchKeys gk(cctx, key.data(), 2);
            std::uint32_t e1 = 0, ec = 0;
            throw_on(sealhip_galois_elt_from_step(cctx.get(), 1, &e1));
            throw_on(sealhip_galois_elt_from_step(cctx.get(), 0, &ec));
            const std::map<std::uint32_t, const KSwitchKeys *> gks{ { e1, &gk }, { ec, &gk } };
            HostCiphertext x = b, cj, rv, rs, rs2;
            x.is_ntt_form() = true;
            cev.complex_conjugate(x, gks, cj);
            cev.rotate_vector(x, 1, gks, rv);
            cev.rescale_to_next(x, rs);
            cev.rescale_to(x, 1, rs2);
            const auto digest = [](const HostCiphertext &ct) {
                std::uint64_t hh = 0xcbf29ce484222325ULL;
                for (std::uint64_t w : ct.words)
                    for (int i = 0; i < 8; i++)
                    {
                        hh ^= (w >> (8 * i)) & 0xff;
                        hh *= 0x100000001b3ULL;
                    }
                return (unsigned long long)hh;
            };
            std::printf("complex_conjugate digest %016llx\n", digest(cj));
            std::printf("rotate_vector digest %016llx\n", digest(rv));
            int logic = 0;
            try { HostCiphertext t = x; cev.rotate_rows_inplace(t, 1, gks); } catch (const std::logic_error &) { logic++; }
            try { HostCiphertext t = x; cev.rotate_columns_inplace(t, gks); } catch (const std::logic_error &) { logic++; }
            const bool ckks_ok = logic == 2 && rs.words == rs2.words && rs.coeff_modulus_size() == 1;
            std::printf("ckks names %s\n", ckks_ok ? "ok" : "FAILED");
            if (!ckks_ok)
                return 1;
        }
    }
    catch (const std::exception &e)
    {
        std::printf("exception: %s\n", e.what());
        return 1;
    }
    return 0;
}
