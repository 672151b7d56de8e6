// evaluator.hpp -- C++ host adapter with seal::Evaluator's method names over the sealhip C ABI.
//
// seal::Evaluator is non-virtual and non-copyable (native/src/seal/evaluator.h:1316-1322), so this is not a
// subclass: it is a class with the same method names and argument meaning for the hot-path operations
// (evaluator.h:246-298 multiply/square, :344-371 relinearize, :396-430 mod_switch_to_next, :479-502 mod_switch_to, :565-583
// rescale_to_next, :606-629 rescale_to, :183 add_many, :902-947 transform_to/from_ntt, :984-1021 apply_galois, :1057-1103
// rotate_rows, :1131-1173 rotate_columns, :1201-1239 rotate_vector, :1269-1308 complex_conjugate, and every
// destination-taking variant), doing
// the same metadata checks on the host and forwarding raw pointers to the ABI. It is a template over the
// ciphertext type so that it compiles both against seal::Ciphertext (where the reference headers exist) and
// against the plain sealhip::HostCiphertext below (everywhere else, e.g. the GPU box).
//
// Required of CT: data() -> uint64_t*, size(), coeff_modulus_size(), poly_modulus_degree(), is_ntt_form()
// (assignable), resize_raw(size, coeff_modulus_size). For seal::Ciphertext the last one is
//   ct.resize(context, parms_id_of_level, size)  -- see INTEGRATION.md.
//
// Exceptions mirror the reference: std::invalid_argument / std::logic_error (evaluator.cpp:238-271).
#pragma once

#include <algorithm>
#include <cstdint>
#include <map>
#include <memory>
#include <stdexcept>
#include <string>
#include <vector>

#include "../../include/sealhip.h"

namespace sealhip_host
{
    inline void throw_on(long hr)
    {
        if (hr == SEALHIP_S_OK)
            return;
        const std::string msg = sealhip_last_error_string();
        if (hr == SEALHIP_E_INVALIDARG || hr == SEALHIP_E_POINTER)
            throw std::invalid_argument(msg);
        if (hr == SEALHIP_COR_E_INVALIDOPERATION)
            throw std::logic_error(msg);
        if (hr == SEALHIP_E_OUTOFMEMORY)
            throw std::bad_alloc();
        throw std::runtime_error(msg);
    }

    // Plain stand-in with the reference's layout (ciphertext.h:359-368): size x k x N uint64, row-major.
    struct HostCiphertext
    {
        std::vector<std::uint64_t> words;
        std::size_t size_ = 0, k_ = 0, n_ = 0;
        bool ntt_form_ = false;
        double scale_ = 1.0;
        std::uint64_t *data() { return words.data(); }
        const std::uint64_t *data() const { return words.data(); }
        std::size_t size() const { return size_; }
        std::size_t coeff_modulus_size() const { return k_; }
        std::size_t poly_modulus_degree() const { return n_; }
        bool &is_ntt_form() { return ntt_form_; }
        bool is_ntt_form() const { return ntt_form_; }
        double &scale() { return scale_; }
        void resize_raw(std::size_t size, std::size_t k)
        {
            // like IntArray::resize: keeps the leading words, zero-fills the rest (ciphertext.cpp:84-133)
            std::vector<std::uint64_t> next(size * k * n_, 0);
            const std::size_t polys = size < size_ ? size : size_;
            const std::size_t rows = k < k_ ? k : k_;
            for (std::size_t s = 0; s < polys; s++)
                for (std::size_t r = 0; r < rows; r++)
                    for (std::size_t c = 0; c < n_; c++)
                        next[(s * k + r) * n_ + c] = words[(s * k_ + r) * n_ + c];
            words.swap(next);
            size_ = size;
            k_ = k;
        }
    };

    class Context
    {
    public:
        explicit Context(const sealhip_params &p) : scheme_(p.scheme), n_(std::size_t(1) << p.log_n)
        {
            throw_on(sealhip_context_create(&p, &ctx_));
        }
        ~Context()
        {
            if (ctx_)
                sealhip_context_destroy(ctx_);
        }
        Context(const Context &) = delete;
        Context &operator=(const Context &) = delete;
        sealhip_context *get() const { return ctx_; }
        std::uint32_t scheme() const { return scheme_; }
        std::size_t n() const { return n_; }

    private:
        sealhip_context *ctx_ = nullptr;
        std::uint32_t scheme_;
        std::size_t n_;
    };

    // device staging of one host object
    class Staged
    {
    public:
        Staged(const Context &c, std::size_t words) : c_(c), words_(words)
        {
            throw_on(sealhip_malloc(c.get(), words * 8, &d_));
        }
        ~Staged()
        {
            if (d_)
                sealhip_free(c_.get(), d_);
        }
        void up(const std::uint64_t *h, std::size_t words) { throw_on(sealhip_memcpy_h2d(c_.get(), d_, h, words * 8)); }
        void down(std::uint64_t *h, std::size_t words) { throw_on(sealhip_memcpy_d2h(c_.get(), h, d_, words * 8)); }
        std::uint64_t *ptr() { return static_cast<std::uint64_t *>(d_); }

    private:
        const Context &c_;
        void *d_ = nullptr;
        std::size_t words_;
    };

    class KSwitchKeys // one key of RelinKeys / GaloisKeys (kswitchkeys.h:92-130), resident on the device
    {
    public:
        KSwitchKeys(const Context &c, const std::uint64_t *host_key, std::uint32_t n_digits) : c_(c)
        {
            throw_on(sealhip_kswitch_key_load(c.get(), host_key, n_digits, 1, &key_));
        }
        ~KSwitchKeys()
        {
            if (key_)
                sealhip_kswitch_key_destroy(c_.get(), key_);
        }
        KSwitchKeys(const KSwitchKeys &) = delete;
        const sealhip_kswitch_key *get() const { return key_; }

    private:
        const Context &c_;
        sealhip_kswitch_key *key_ = nullptr;
    };

    template <class CT>
    class Evaluator
    {
    public:
        explicit Evaluator(const Context &context) : ctx_(context) {}

        // Evaluator::multiply_inplace (evaluator.cpp:235-272)
        void multiply_inplace(CT &encrypted1, const CT &encrypted2)
        {
            check_pair(encrypted1, encrypted2);
            const bool bfv = ctx_.scheme() == SEALHIP_SCHEME_BFV;
            if (bfv && (encrypted1.is_ntt_form() || encrypted2.is_ntt_form()))
                throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form"); // :276-279
            if (!bfv && !(encrypted1.is_ntt_form() && encrypted2.is_ntt_form()))
                throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form"); // :449-452
            const std::size_t k = encrypted1.coeff_modulus_size(), n = ctx_.n();
            const std::size_t s1 = encrypted1.size(), s2 = encrypted2.size(), dest = s1 + s2 - 1;
            Staged a(ctx_, s1 * k * n), b(ctx_, s2 * k * n), o(ctx_, dest * k * n);
            a.up(encrypted1.data(), s1 * k * n);
            b.up(encrypted2.data(), s2 * k * n);
            throw_on(sealhip_evaluator_multiply(ctx_.get(), std::uint32_t(k), a.ptr(), std::uint32_t(s1), b.ptr(),
                                                std::uint32_t(s2), 1, o.ptr()));
            encrypted1.resize_raw(dest, k); // :324 / :484
            o.down(encrypted1.data(), dest * k * n);
        }

        void square_inplace(CT &encrypted)
        {
            const CT copy = encrypted;
            multiply_inplace(encrypted, copy); // same canonical residues as bfv_square / ckks_square (:560-770)
        }

        // Evaluator::relinearize_inplace (evaluator.cpp:772-827); relin_keys[i] = key of get_index(i + 2)
        void relinearize_inplace(CT &encrypted, const std::vector<const KSwitchKeys *> &relin_keys)
        {
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n(), size = encrypted.size();
            if (size < 2)
                throw std::invalid_argument("encrypted is not valid for encryption parameters");
            if (relin_keys.size() + 2 < size)
                throw std::invalid_argument("not enough relinearization keys"); // :793-796
            if (size == 2)
                return;
            std::vector<const sealhip_kswitch_key *> raw;
            for (auto *rk : relin_keys)
                raw.push_back(rk ? rk->get() : nullptr);
            Staged c(ctx_, size * k * n);
            c.up(encrypted.data(), size * k * n);
            throw_on(sealhip_evaluator_relinearize(ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1,
                                                   raw.data(), std::uint32_t(raw.size())));
            c.down(encrypted.data(), size * k * n);
            encrypted.resize_raw(2, k); // :819
        }

        // Evaluator::mod_switch_to_next_inplace (evaluator.cpp:996-1036)
        void mod_switch_to_next_inplace(CT &encrypted) { switch_level(encrypted, false); }
        // Evaluator::rescale_to_next_inplace (evaluator.cpp:1090-1126); the caller updates scale() /= q_last (:889-890)
        void rescale_to_next_inplace(CT &encrypted) { switch_level(encrypted, true); }

        void transform_to_ntt_inplace(CT &encrypted)
        {
            if (encrypted.is_ntt_form())
                throw std::invalid_argument("encrypted is already in NTT form"); // :1759-1762
            transform(encrypted, true);
            encrypted.is_ntt_form() = true;
        }
        void transform_from_ntt_inplace(CT &encrypted_ntt)
        {
            if (!encrypted_ntt.is_ntt_form())
                throw std::invalid_argument("encrypted_ntt is not in NTT form"); // :1807-1810
            transform(encrypted_ntt, false);
            encrypted_ntt.is_ntt_form() = false;
        }

        // Evaluator::apply_galois_inplace (evaluator.cpp:1841-1943)
        void apply_galois_inplace(CT &encrypted, std::uint32_t galois_elt, const KSwitchKeys &galois_key)
        {
            if (encrypted.size() > 2)
                throw std::invalid_argument("encrypted size must be 2"); // :1884-1887
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n();
            Staged c(ctx_, 2 * k * n);
            c.up(encrypted.data(), 2 * k * n);
            throw_on(sealhip_evaluator_apply_galois(ctx_.get(), std::uint32_t(k), c.ptr(), 1, galois_elt,
                                                    galois_key.get()));
            c.down(encrypted.data(), 2 * k * n);
        }

        // Evaluator::rotate_vector_inplace (evaluator.h:1201-1211): CKKS only, then rotate_internal
        void rotate_vector_inplace(CT &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (ctx_.scheme() != SEALHIP_SCHEME_CKKS)
                throw std::logic_error("unsupported scheme"); // :1205-1208
            rotate_vector_like(encrypted, steps, galois_keys);
        }
        // rotate_internal (evaluator.cpp:1945-2000): one automorphism when its key is present, else the NAF of the step count
        void rotate_vector_like(CT &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (steps == 0)
                return;
            std::uint32_t elt = 0;
            throw_on(sealhip_galois_elt_from_step(ctx_.get(), steps, &elt));
            auto it = galois_keys.find(elt);
            if (it != galois_keys.end())
                return apply_galois_inplace(encrypted, elt, *it->second);
            std::vector<int> naf; // util/numth.h:22-42
            bool neg = steps < 0;
            int v = neg ? -steps : steps;
            for (int i = 0; v; i++)
            {
                int zi = (v % 2) ? 2 - (v % 4) : 0;
                v = (v - zi) / 2;
                if (zi)
                    naf.push_back((neg ? -zi : zi) * (1 << i));
            }
            if (naf.size() == 1)
                throw std::invalid_argument("Galois key not present"); // :1985-1988
            for (int s : naf)
                if (std::size_t(s < 0 ? -s : s) != (ctx_.n() >> 1))
                    rotate_vector_like(encrypted, s, galois_keys);
        }

        // Evaluator::rotate_rows_inplace (evaluator.h:1057-1067): BFV only, then rotate_internal like rotate_vector
        void rotate_rows_inplace(CT &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (ctx_.scheme() != SEALHIP_SCHEME_BFV)
                throw std::logic_error("unsupported scheme"); // :1061-1064
            rotate_vector_like(encrypted, steps, galois_keys);
        }
        // Evaluator::rotate_columns_inplace (evaluator.h:1131-1139): BFV only; conjugate_internal = apply_galois with
        // get_elt_from_step(0) = 2N - 1 (:1343-1363)
        void rotate_columns_inplace(CT &encrypted, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (ctx_.scheme() != SEALHIP_SCHEME_BFV)
                throw std::logic_error("unsupported scheme"); // :1134-1137
            conjugate_internal(encrypted, galois_keys);
        }
        // Evaluator::complex_conjugate_inplace (evaluator.h:1269-1277): CKKS only, the same automorphism
        void complex_conjugate_inplace(CT &encrypted, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (ctx_.scheme() != SEALHIP_SCHEME_CKKS)
                throw std::logic_error("unsupported scheme"); // :1272-1275
            conjugate_internal(encrypted, galois_keys);
        }

        // Evaluator::mod_switch_to_inplace (evaluator.cpp:1038-1060) / rescale_to_inplace (:1128-1165). The ABI names a level by
        // its number of primes k (the chain drops one prime per level, context.cpp:423-431); for seal::Ciphertext the
        // binding maps the parms_id argument to it.
        void mod_switch_to_inplace(CT &encrypted, std::size_t target_coeff_modulus_size)
        {
            if (target_coeff_modulus_size < 1)
                throw std::invalid_argument("parms_id is not valid for encryption parameters"); // :1047-1050
            if (encrypted.coeff_modulus_size() < target_coeff_modulus_size)
                throw std::invalid_argument("cannot switch to higher level modulus"); // :1051-1054
            while (encrypted.coeff_modulus_size() != target_coeff_modulus_size)
                mod_switch_to_next_inplace(encrypted); // :1056-1059
        }
        void rescale_to_inplace(CT &encrypted, std::size_t target_coeff_modulus_size)
        {
            if (target_coeff_modulus_size < 1)
                throw std::invalid_argument("parms_id is not valid for encryption parameters"); // :1137-1140
            if (encrypted.coeff_modulus_size() < target_coeff_modulus_size)
                throw std::invalid_argument("cannot switch to higher level modulus"); // :1141-1144
            if (ctx_.scheme() != SEALHIP_SCHEME_CKKS)
                throw std::invalid_argument("unsupported operation for scheme type"); // :1146-1162
            while (encrypted.coeff_modulus_size() != target_coeff_modulus_size)
                rescale_to_next_inplace(encrypted);
        }
        // Evaluator::add_many (evaluator.cpp:153-172)
        void add_many(const std::vector<CT> &encrypteds, CT &destination)
        {
            if (encrypteds.empty())
                throw std::invalid_argument("encrypteds cannot be empty"); // :155-158
            for (const CT &c : encrypteds)
                if (&c == &destination)
                    throw std::invalid_argument("encrypteds must be different from destination"); // :159-165
            destination = encrypteds[0];
            for (std::size_t i = 1; i < encrypteds.size(); i++)
                add_inplace(destination, encrypteds[i]);
        }

        // ---- destination-taking variants (evaluator.h:121-126, :156-168, :214-226, :268-284, :317-322, :371-376, :396-430,
        // :565-583, :916-947, :1021-1027, :1097-1103, :1167-1173, :1239-1245, :1302-1308): copy, then the in-place form
        void negate(const CT &encrypted, CT &destination) { destination = encrypted; negate_inplace(destination); }
        void add(const CT &encrypted1, const CT &encrypted2, CT &destination)
        {
            if (&encrypted2 == &destination) // (:160-163: addition commutes, the alias is kept valid)
                add_inplace(destination, encrypted1);
            else
            {
                destination = encrypted1;
                add_inplace(destination, encrypted2);
            }
        }
        void sub(const CT &encrypted1, const CT &encrypted2, CT &destination)
        {
            if (&encrypted2 == &destination) // :216-222: destination = -(encrypted2 - encrypted1)
            {
                sub_inplace(destination, encrypted1);
                negate_inplace(destination);
            }
            else
            {
                destination = encrypted1;
                sub_inplace(destination, encrypted2);
            }
        }
        void multiply(const CT &encrypted1, const CT &encrypted2, CT &destination)
        {
            if (&encrypted2 == &destination) // :272-275
                multiply_inplace(destination, encrypted1);
            else
            {
                destination = encrypted1;
                multiply_inplace(destination, encrypted2);
            }
        }
        void square(const CT &encrypted, CT &destination) { destination = encrypted; square_inplace(destination); }
        void relinearize(const CT &encrypted, const std::vector<const KSwitchKeys *> &relin_keys, CT &destination)
        {
            destination = encrypted;
            relinearize_inplace(destination, relin_keys);
        }
        void mod_switch_to_next(const CT &encrypted, CT &destination) { destination = encrypted; mod_switch_to_next_inplace(destination); }
        void rescale_to_next(const CT &encrypted, CT &destination) { destination = encrypted; rescale_to_next_inplace(destination); }
        void mod_switch_to(const CT &encrypted, std::size_t target_coeff_modulus_size, CT &destination)
        {
            destination = encrypted;
            mod_switch_to_inplace(destination, target_coeff_modulus_size);
        }
        void rescale_to(const CT &encrypted, std::size_t target_coeff_modulus_size, CT &destination)
        {
            destination = encrypted;
            rescale_to_inplace(destination, target_coeff_modulus_size);
        }
        void transform_to_ntt(const CT &encrypted, CT &destination_ntt) { destination_ntt = encrypted; transform_to_ntt_inplace(destination_ntt); }
        void transform_from_ntt(const CT &encrypted_ntt, CT &destination) { destination = encrypted_ntt; transform_from_ntt_inplace(destination); }
        void apply_galois(const CT &encrypted, std::uint32_t galois_elt, const KSwitchKeys &galois_key, CT &destination)
        {
            destination = encrypted;
            apply_galois_inplace(destination, galois_elt, galois_key);
        }
        void rotate_vector(const CT &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys, CT &destination)
        {
            destination = encrypted;
            rotate_vector_inplace(destination, steps, galois_keys);
        }
        void rotate_rows(const CT &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys, CT &destination)
        {
            destination = encrypted;
            rotate_rows_inplace(destination, steps, galois_keys);
        }
        void rotate_columns(const CT &encrypted, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys, CT &destination)
        {
            destination = encrypted;
            rotate_columns_inplace(destination, galois_keys);
        }
        void complex_conjugate(const CT &encrypted, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys, CT &destination)
        {
            destination = encrypted;
            complex_conjugate_inplace(destination, galois_keys);
        }
        void multiply_plain(const CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form, CT &destination)
        {
            destination = encrypted;
            multiply_plain_inplace(destination, plain, plain_is_ntt_form);
        }
        void add_plain(const CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form, CT &destination)
        {
            destination = encrypted;
            add_plain_inplace(destination, plain, plain_is_ntt_form);
        }
        void sub_plain(const CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form, CT &destination)
        {
            destination = encrypted;
            sub_plain_inplace(destination, plain, plain_is_ntt_form);
        }
        void exponentiate(const CT &encrypted, std::uint64_t exponent, const std::vector<const KSwitchKeys *> &relin_keys, CT &destination)
        {
            destination = encrypted; // :719-726
            exponentiate_inplace(destination, exponent, relin_keys);
        }

        // Evaluator::multiply_many (evaluator.cpp:1180-1255): destination = product of all, relinearized after every step
        void multiply_many(const std::vector<CT> &encrypteds, const std::vector<const KSwitchKeys *> &relin_keys, CT &destination)
        {
            if (encrypteds.empty())
                throw std::invalid_argument("encrypteds vector must not be empty"); // :1185-1188
            for (const CT &c : encrypteds)
                if (&c == &destination)
                    throw std::invalid_argument("encrypteds must be different from destination"); // :1193-1199
            const std::size_t k = encrypteds[0].coeff_modulus_size(), n = ctx_.n(), words = 2 * k * n;
            std::vector<std::unique_ptr<Staged>> dev;
            std::vector<const std::uint64_t *> ptrs;
            for (const CT &c : encrypteds)
            {
                if (c.size() != 2 || c.coeff_modulus_size() != k || c.poly_modulus_degree() != n)
                    throw std::invalid_argument("encrypteds is not valid for encryption parameters");
                dev.emplace_back(new Staged(ctx_, words));
                dev.back()->up(c.data(), words);
                ptrs.push_back(dev.back()->ptr());
            }
            std::vector<const sealhip_kswitch_key *> raw;
            for (auto *rk : relin_keys)
                raw.push_back(rk ? rk->get() : nullptr);
            Staged o(ctx_, words);
            throw_on(sealhip_evaluator_multiply_many(ctx_.get(), std::uint32_t(k), ptrs.data(), std::uint32_t(ptrs.size()), 1,
                                                     raw.data(), std::uint32_t(raw.size()), o.ptr()));
            destination = encrypteds[0];
            destination.resize_raw(2, k);
            o.down(destination.data(), words);
        }
        // Evaluator::exponentiate_inplace (evaluator.cpp:1257-1288)
        void exponentiate_inplace(CT &encrypted, std::uint64_t exponent, const std::vector<const KSwitchKeys *> &relin_keys)
        {
            if (exponent == 0)
                throw std::invalid_argument("exponent cannot be 0"); // :1275-1278
            if (exponent == 1)
                return; // :1281-1284
            const std::vector<CT> copies(static_cast<std::size_t>(exponent), encrypted);
            multiply_many(copies, relin_keys, encrypted);
        }

        // ---- batches: what `for (auto &ct : cts) evaluator.multiply_inplace(ct, other)` does in the reference, as one call
        // that pipelines the separately allocated ciphertexts through the device (sealhip_evaluator_multiply_host).
        // encrypted1[i] *= encrypted2[i]; with relin_keys the products come back relinearized (size 2).
        void multiply_inplace(std::vector<CT *> &encrypted1, const std::vector<const CT *> &encrypted2,
                              const std::vector<const KSwitchKeys *> *relin_keys = nullptr)
        {
            const std::size_t count = encrypted1.size();
            if (count != encrypted2.size())
                throw std::invalid_argument("encrypted1 and encrypted2 batch size mismatch");
            if (!count)
                return;
            const std::size_t k = encrypted1[0]->coeff_modulus_size(), s1 = encrypted1[0]->size(), s2 = encrypted2[0]->size();
            const bool bfv = ctx_.scheme() == SEALHIP_SCHEME_BFV;
            for (std::size_t i = 0; i < count; i++)
            {
                check_pair(*encrypted1[i], *encrypted2[i]);
                if (encrypted1[i]->coeff_modulus_size() != k || encrypted1[i]->size() != s1 || encrypted2[i]->size() != s2)
                    throw std::invalid_argument("a batch must be uniform in level and size");
                if (bfv && (encrypted1[i]->is_ntt_form() || encrypted2[i]->is_ntt_form()))
                    throw std::invalid_argument("encrypted1 or encrypted2 cannot be in NTT form");
                if (!bfv && !(encrypted1[i]->is_ntt_form() && encrypted2[i]->is_ntt_form()))
                    throw std::invalid_argument("encrypted1 or encrypted2 must be in NTT form");
            }
            const std::size_t dest = s1 + s2 - 1, out_size = relin_keys && dest > 2 ? 2 : dest;
            std::vector<const sealhip_kswitch_key *> raw;
            if (relin_keys)
                for (auto *rk : *relin_keys)
                    raw.push_back(rk ? rk->get() : nullptr);
            // the results land in fresh buffers of the final size, then replace the operands' storage
            std::vector<std::vector<std::uint64_t>> res(count, std::vector<std::uint64_t>(out_size * k * ctx_.n()));
            std::vector<const std::uint64_t *> pa(count), pb(count);
            std::vector<std::uint64_t *> po(count);
            for (std::size_t i = 0; i < count; i++)
            {
                pa[i] = encrypted1[i]->data();
                pb[i] = encrypted2[i]->data();
                po[i] = res[i].data();
            }
            throw_on(sealhip_evaluator_multiply_host(ctx_.get(), std::uint32_t(k), pa.data(), std::uint32_t(s1), pb.data(),
                                                     std::uint32_t(s2), count, po.data(), relin_keys ? raw.data() : nullptr,
                                                     std::uint32_t(raw.size())));
            for (std::size_t i = 0; i < count; i++)
            {
                encrypted1[i]->resize_raw(out_size, k);
                std::copy(res[i].begin(), res[i].end(), encrypted1[i]->data());
            }
        }
        void rotate_vector_inplace(std::vector<CT *> &encrypted, int steps, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            if (encrypted.empty() || steps == 0)
                return;
            const std::size_t k = encrypted[0]->coeff_modulus_size();
            std::vector<std::uint64_t *> p;
            for (CT *ct : encrypted)
            {
                if (ct->size() != 2 || ct->coeff_modulus_size() != k)
                    throw std::invalid_argument("encrypted size must be 2"); // :1884-1887
                p.push_back(ct->data());
            }
            std::vector<std::uint32_t> elts;
            std::vector<const sealhip_kswitch_key *> keys;
            for (auto &kv : galois_keys)
            {
                elts.push_back(kv.first);
                keys.push_back(kv.second ? kv.second->get() : nullptr);
            }
            throw_on(sealhip_evaluator_rotate_vector_host(ctx_.get(), std::uint32_t(k), p.data(), p.size(), steps, elts.data(),
                                                          keys.data(), std::uint32_t(elts.size())));
        }

        // The vector entries above read / write the ciphertexts' own buffers; a pool block (mempool.cpp:45,145) pinned here
        // once lets the DMA engine do that in place (sealhip_host_register). Unregister before freeing.
        void register_pool_block(void *ptr, std::size_t bytes)
        {
            throw_on(sealhip_host_register(ctx_.get(), ptr, bytes));
        }
        void unregister_pool_block(void *ptr)
        {
            throw_on(sealhip_host_unregister(ctx_.get(), ptr));
        }

        // ---- the rest of the Evaluator surface
        // Evaluator::negate_inplace (evaluator.cpp:65-88)
        void negate_inplace(CT &encrypted)
        {
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n(), size = encrypted.size();
            Staged c(ctx_, size * k * n);
            c.up(encrypted.data(), size * k * n);
            throw_on(sealhip_evaluator_negate(ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1, c.ptr()));
            c.down(encrypted.data(), size * k * n);
        }
        // Evaluator::add_inplace (evaluator.cpp:90-151) / sub_inplace (:174-233)
        void add_inplace(CT &encrypted1, const CT &encrypted2) { add_sub(encrypted1, encrypted2, false); }
        void sub_inplace(CT &encrypted1, const CT &encrypted2) { add_sub(encrypted1, encrypted2, true); }
        // Evaluator::multiply_plain_inplace (evaluator.cpp:1438-1473). plain: NTT form -> k*N words (multiply_plain_ntt,
        // :1605-1646), coefficient form -> N coefficients below t (multiply_plain_normal, :1475-1603).
        void multiply_plain_inplace(CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form)
        {
            if (encrypted.is_ntt_form() != plain_is_ntt_form)
                throw std::invalid_argument("NTT form mismatch"); // :1449-1452
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n(), size = encrypted.size();
            const std::size_t pw = plain_is_ntt_form ? k * n : n;
            Staged c(ctx_, size * k * n), p(ctx_, pw);
            c.up(encrypted.data(), size * k * n);
            p.up(plain, pw);
            throw_on((plain_is_ntt_form ? sealhip_evaluator_multiply_plain_ntt : sealhip_evaluator_multiply_plain)(
                ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1, p.ptr(), 0));
            c.down(encrypted.data(), size * k * n);
        }
        // Evaluator::add_plain_inplace (evaluator.cpp:1290-1362) / sub_plain_inplace (:1364-1435). BFV: plain = N
        // coefficients below t, ciphertext in coefficient form; CKKS: plain = k*N words in NTT form, same level.
        void add_plain_inplace(CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form)
        {
            plain_linear(encrypted, plain, plain_is_ntt_form, false);
        }
        void sub_plain_inplace(CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form)
        {
            plain_linear(encrypted, plain, plain_is_ntt_form, true);
        }
        // Ciphertext::is_transparent (ciphertext.h:471-476) evaluated on the device copy
        bool is_transparent(const CT &encrypted)
        {
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n(), size = encrypted.size();
            Staged c(ctx_, size * k * n);
            c.up(encrypted.data(), size * k * n);
            std::uint8_t flag = 0;
            throw_on(sealhip_is_transparent(ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1, &flag));
            return flag != 0;
        }

    private:
        // conjugate_internal (evaluator.h:1343-1363): the automorphism x -> x^(2N-1), i.e. get_elt_from_step(0)
        void conjugate_internal(CT &encrypted, const std::map<std::uint32_t, const KSwitchKeys *> &galois_keys)
        {
            std::uint32_t elt = 0;
            throw_on(sealhip_galois_elt_from_step(ctx_.get(), 0, &elt));
            auto it = galois_keys.find(elt);
            if (it == galois_keys.end() || !it->second)
                throw std::invalid_argument("Galois key not present"); // evaluator.cpp:1871-1874
            apply_galois_inplace(encrypted, elt, *it->second);
        }
        void plain_linear(CT &encrypted, const std::uint64_t *plain, bool plain_is_ntt_form, bool sub)
        {
            if (ctx_.scheme() == SEALHIP_SCHEME_BFV && encrypted.is_ntt_form())
                throw std::invalid_argument("BFV encrypted cannot be in NTT form"); // :1304-1307
            if (ctx_.scheme() == SEALHIP_SCHEME_CKKS && !encrypted.is_ntt_form())
                throw std::invalid_argument("CKKS encrypted must be in NTT form"); // :1308-1311
            if (encrypted.is_ntt_form() != plain_is_ntt_form)
                throw std::invalid_argument("NTT form mismatch"); // :1312-1315
            const std::size_t k = encrypted.coeff_modulus_size(), n = ctx_.n(), size = encrypted.size();
            const std::size_t pw = plain_is_ntt_form ? k * n : n;
            Staged c(ctx_, size * k * n), p(ctx_, pw);
            c.up(encrypted.data(), size * k * n);
            p.up(plain, pw);
            throw_on(sealhip_evaluator_add_plain(ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1, p.ptr(), pw,
                                                 sub ? 1 : 0));
            c.down(encrypted.data(), size * k * n);
        }
        void add_sub(CT &a, const CT &b, bool sub)
        {
            if (a.poly_modulus_degree() != ctx_.n() || b.poly_modulus_degree() != ctx_.n() || a.size() < 1 || b.size() < 1)
                throw std::invalid_argument("encrypted1 is not valid for encryption parameters"); // :93-100
            if (a.coeff_modulus_size() != b.coeff_modulus_size())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch"); // :101-104
            if (a.is_ntt_form() != b.is_ntt_form())
                throw std::invalid_argument("NTT form mismatch"); // :105-108
            const std::size_t k = a.coeff_modulus_size(), n = ctx_.n(), sa = a.size(), sb = b.size();
            const std::size_t so = sa > sb ? sa : sb;
            Staged x(ctx_, sa * k * n), y(ctx_, sb * k * n), o(ctx_, so * k * n);
            x.up(a.data(), sa * k * n);
            y.up(b.data(), sb * k * n);
            throw_on((sub ? sealhip_evaluator_sub : sealhip_evaluator_add)(ctx_.get(), std::uint32_t(k), x.ptr(),
                                                                           std::uint32_t(sa), y.ptr(), std::uint32_t(sb), 1,
                                                                           o.ptr()));
            a.resize_raw(so, k); // :131-132
            o.down(a.data(), so * k * n);
        }
        void check_pair(const CT &a, const CT &b) const
        {
            if (a.poly_modulus_degree() != ctx_.n() || b.poly_modulus_degree() != ctx_.n() || a.size() < 2 || b.size() < 2)
                throw std::invalid_argument("encrypted1 is not valid for encryption parameters"); // :238-245
            if (a.coeff_modulus_size() != b.coeff_modulus_size())
                throw std::invalid_argument("encrypted1 and encrypted2 parameter mismatch"); // :246-249
        }
        void switch_level(CT &ct, bool rescale)
        {
            const std::size_t k = ct.coeff_modulus_size(), n = ctx_.n(), size = ct.size();
            if (k < 2)
                throw std::invalid_argument("end of modulus switching chain reached"); // :1005-1008
            Staged c(ctx_, size * k * n), o(ctx_, size * (k - 1) * n);
            c.up(ct.data(), size * k * n);
            throw_on((rescale ? sealhip_evaluator_rescale_to_next : sealhip_evaluator_mod_switch_to_next)(
                ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1, o.ptr()));
            ct.resize_raw(size, k - 1); // :879
            o.down(ct.data(), size * (k - 1) * n);
        }
        void transform(CT &ct, bool to_ntt)
        {
            const std::size_t k = ct.coeff_modulus_size(), n = ctx_.n(), size = ct.size();
            Staged c(ctx_, size * k * n);
            c.up(ct.data(), size * k * n);
            throw_on((to_ntt ? sealhip_evaluator_transform_to_ntt : sealhip_evaluator_transform_from_ntt)(
                ctx_.get(), std::uint32_t(k), c.ptr(), std::uint32_t(size), 1));
            c.down(ct.data(), size * k * n);
        }
        const Context &ctx_;
    };
} // namespace sealhip_host
