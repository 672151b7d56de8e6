// wire.cpp -- the reference's ciphertext wire format, uncompressed mode:
//   Serialization::SEALHeader        native/src/seal/serialization.h:69-90   (16 bytes, magic 0xA15E, version 3.5)
//   Ciphertext::save_members         native/src/seal/ciphertext.cpp:170-226
//   IntArray<T>::save_members        native/src/seal/intarray.h:592-620      (nested SEALHeader + count + words)
// Parsing happens on the host; the coefficient words go straight from the caller's buffer to HBM (and back),
// so the host library never touches coefficient data. The reference build has no zlib, so
// compr_mode_type::deflate is rejected exactly as Serialization::IsValidHeader does without SEAL_USE_ZLIB.
#include "../../include/sealhip.h"

#include <cstring>

#include "blake2xb.hpp"
#include "engine.hpp"

namespace sealhip
{
    namespace
    {
        constexpr std::uint16_t kMagic = 0xA15E;    // serialization.h:51
        constexpr std::uint8_t kHeaderSize = 0x10;  // serialization.h:56
        constexpr std::uint8_t kVersionMajor = 3, kVersionMinor = 5; // CMakeLists.txt:14 (SEAL 3.5.3)
        constexpr std::size_t kMembersFixed = 32 + 1 + 8 + 8 + 8 + 8; // parms_id, is_ntt_form, size, N, k, scale
        constexpr std::size_t kSeedBytes = 64;                        // random_seed_type, randomgen.h (8 x uint64)
        constexpr std::uint64_t kSeedMarker = 0xFFFFFFFFFFFFFFFFULL;  // ciphertext.h:704-707

        struct Header
        {
            std::uint16_t magic;
            std::uint8_t header_size, version_major, version_minor, compr_mode;
            std::uint16_t reserved;
            std::uint64_t size;
        };
        static_assert(sizeof(Header) == 16, "SEALHeader is 16 bytes (tests/seal/serialization.cpp:50)");

        bool header_valid(const Header &h) // Serialization::IsValidHeader, serialization.h:140-162 (no zlib)
        {
            return h.magic == kMagic && h.header_size == kHeaderSize && h.version_major == kVersionMajor &&
                   h.version_minor == kVersionMinor && h.compr_mode == 0;
        }

        // Serialization::LoadHeader with try_upgrade_if_invalid (serialization.cpp:137-165): a SEAL 3.4 header
        // {magic u16, zero u8, compr_mode u8, size u32, reserved u64} is upgraded in place
        Header read_header(const unsigned char *p, std::size_t len)
        {
            if (len < sizeof(Header))
                throw std::runtime_error("I/O error");
            Header h;
            std::memcpy(&h, p, sizeof(h));
            if (!header_valid(h))
            {
                std::uint32_t size32;
                std::memcpy(&size32, p + 4, 4);
                Header up{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, p[3], 0, size32 };
                if (header_valid(up))
                    h = up;
            }
            if (h.version_major != kVersionMajor || h.version_minor != kVersionMinor)
                throw std::logic_error("incompatible version"); // serialization.cpp:345-348
            if (!header_valid(h))
                throw std::logic_error("loaded SEALHeader is invalid"); // :349-352
            return h;
        }

        struct Parsed
        {
            sealhip_ciphertext_info info;
            const unsigned char *words; // start of the coefficient words
        };

        Parsed parse(const void *bytes, std::size_t len)
        {
            const unsigned char *p = static_cast<const unsigned char *>(bytes);
            const Header outer = read_header(p, len);
            if (outer.size > len || outer.size < sizeof(Header) + kMembersFixed + sizeof(Header) + 8)
                throw std::runtime_error("I/O error");
            const unsigned char *end = p + outer.size;
            p += sizeof(Header);
            Parsed out{};
            sealhip_ciphertext_info &ci = out.info;
            std::memcpy(ci.parms_id, p, 32); // ciphertext.cpp:248-259
            p += 32;
            ci.is_ntt_form = *p++ != 0;
            std::uint64_t size64, n64, k64;
            std::memcpy(&size64, p, 8);
            std::memcpy(&n64, p + 8, 8);
            std::memcpy(&k64, p + 16, 8);
            std::memcpy(&ci.scale, p + 24, 8);
            p += 32;
            if (size64 > 0xFFFFFFFFull || k64 > 0xFFFFFFFFull)
                throw std::logic_error("ciphertext data is invalid");
            ci.size = static_cast<std::uint32_t>(size64);
            ci.coeff_modulus_size = static_cast<std::uint32_t>(k64);
            ci.poly_modulus_degree = n64;
            // nested IntArray (intarray.h:622-650) behind its own header
            const Header inner = read_header(p, static_cast<std::size_t>(end - p));
            if (inner.size > static_cast<std::size_t>(end - p) || inner.size < sizeof(Header) + 8)
                throw std::runtime_error("I/O error");
            p += sizeof(Header);
            std::uint64_t count;
            std::memcpy(&count, p, 8);
            p += 8;
            if (inner.size != sizeof(Header) + 8 + count * 8 || count > (static_cast<std::uint64_t>(1) << 40))
                throw std::runtime_error("I/O error");
            ci.data_words = count;
            out.words = p;
            p += count * 8;
            ci.seeded = static_cast<std::size_t>(end - p) == kSeedBytes ? 1u : 0u; // ciphertext.cpp:296-309
            if (static_cast<std::size_t>(end - p) != (ci.seeded ? kSeedBytes : 0))
                throw std::runtime_error("I/O error");
            ci.total_bytes = outer.size;
            return out;
        }

        // is_metadata_valid_for(ciphertext, context, allow_pure_key_levels = true), valcheck.cpp:67-105
        int level_of(const Engine &e, const sealhip_ciphertext_info &ci)
        {
            int k = -1;
            {
                std::lock_guard<std::mutex> lock(e.mu); // wire_set_parms_id may run on another thread
                for (const auto &kv : e.parms_ids)
                    if (std::memcmp(kv.second.data(), ci.parms_id, 32) == 0)
                        k = kv.first;
            }
            if (k < 0)
                throw std::logic_error("ciphertext data is invalid"); // no ContextData for the parms_id (:76-80)
            if (static_cast<int>(ci.coeff_modulus_size) != k || ci.poly_modulus_degree != e.n)
                throw std::logic_error("ciphertext data is invalid"); // :90-95
            if ((ci.size < 2 && ci.size != 0) || ci.size > 16)
                throw std::logic_error("ciphertext data is invalid"); // :97-102, SEAL_CIPHERTEXT_SIZE_MIN/MAX
            return k;
        }
    } // namespace

    // Ciphertext::expand_seed (ciphertext.cpp:126-133): the rows x N words of c_1 from the 64-byte seed that follows the
    // single stored polynomial of a seeded ciphertext (:296-309). Host work (BLAKE2Xb, sequential rejection sampling).
    std::vector<u64> wire_expand_seed(const Engine &e, int rows, const unsigned char *seed_bytes)
    {
        std::uint64_t seed[8];
        std::memcpy(seed, seed_bytes, sizeof(seed));
        BlakePrng prng(seed);
        std::vector<u64> c1(static_cast<std::size_t>(rows) * e.n);
        static_assert(sizeof(u64) == sizeof(std::uint64_t), "word size");
        sample_poly_uniform(prng, reinterpret_cast<const std::uint64_t *>(e.key_moduli.data()), static_cast<std::size_t>(rows),
                            e.n, reinterpret_cast<std::uint64_t *>(c1.data()));
        return c1;
    }

    void wire_set_parms_id(Engine &e, int k, const std::uint64_t *id)
    {
        if (k < 1 || k > e.n_key)
            throw std::invalid_argument("level k out of range");
        std::array<std::uint64_t, 4> a{ id[0], id[1], id[2], id[3] };
        std::lock_guard<std::mutex> lock(e.mu);
        e.parms_ids[k] = a;
    }

    void wire_peek(const void *bytes, std::size_t len, sealhip_ciphertext_info *info)
    {
        *info = parse(bytes, len).info;
    }

    // Ciphertext::load (ciphertext.cpp:228-330) with the words landing in HBM
    void wire_load(Engine &e, const void *bytes, std::size_t len, sealhip_ciphertext_info *info, u64 *dst,
                   std::size_t capacity_words)
    {
        const Parsed ps = parse(bytes, len);
        *info = ps.info;
        const int k = level_of(e, ps.info);
        const std::uint64_t total = static_cast<std::uint64_t>(ps.info.size) * e.n * k;
        if (ps.info.data_words > total)
            throw std::logic_error("unexpected size"); // intarray.h:633-638
        if (total > capacity_words)
            throw std::invalid_argument("destination buffer is too small");
        if (ps.info.seeded)
        {
            // one stored polynomial + the seed: c_0 comes from the stream, c_1 is re-sampled from the seed
            // (ciphertext.cpp:296-309 -> expand_seed :126-133)
            const std::size_t half = static_cast<std::size_t>(k) * e.n;
            if (ps.info.size != 2 || ps.info.data_words != half)
                throw std::logic_error("ciphertext data is invalid");
            const std::vector<u64> c1 = wire_expand_seed(e, k, ps.words + half * 8);
            SEALHIP_CHECK(hipMemcpyAsync(dst, ps.words, half * 8, hipMemcpyHostToDevice, e.lane().stream));
            SEALHIP_CHECK(hipMemcpyAsync(dst + half, c1.data(), half * 8, hipMemcpyHostToDevice, e.lane().stream));
            SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream));
            return;
        }
        if (ps.info.data_words != total)
            throw std::logic_error("ciphertext data is invalid"); // is_buffer_valid, valcheck.cpp:228-240
        if (total)
            SEALHIP_CHECK(hipMemcpyAsync(dst, ps.words, total * 8, hipMemcpyHostToDevice, e.lane().stream));
        SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream)); // the caller's buffer may go away after the call
    }

    std::size_t wire_save_size(std::uint32_t size, std::uint32_t k, std::size_t n)
    {
        // Ciphertext::save_size(compr_mode_type::none), ciphertext.cpp:135-168
        return sizeof(Header) + kMembersFixed + sizeof(Header) + 8 + static_cast<std::size_t>(size) * k * n * 8;
    }

    // Ciphertext::save (ciphertext.cpp:170-226), unseeded
    std::size_t wire_save(Engine &e, const sealhip_ciphertext_info &ci, const u64 *src, void *bytes, std::size_t capacity)
    {
        const int k = level_of(e, ci);
        const std::size_t words = static_cast<std::size_t>(ci.size) * k * e.n;
        const std::size_t total = wire_save_size(ci.size, static_cast<std::uint32_t>(k), e.n);
        if (capacity < total)
            throw std::invalid_argument("destination buffer is too small");
        unsigned char *p = static_cast<unsigned char *>(bytes);
        const Header outer{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, total };
        std::memcpy(p, &outer, sizeof(outer));
        p += sizeof(outer);
        std::memcpy(p, ci.parms_id, 32);
        p += 32;
        *p++ = ci.is_ntt_form ? 1 : 0;
        const std::uint64_t size64 = ci.size, n64 = e.n, k64 = static_cast<std::uint64_t>(k);
        std::memcpy(p, &size64, 8);
        std::memcpy(p + 8, &n64, 8);
        std::memcpy(p + 16, &k64, 8);
        std::memcpy(p + 24, &ci.scale, 8);
        p += 32;
        const Header inner{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, sizeof(Header) + 8 + words * 8 };
        std::memcpy(p, &inner, sizeof(inner));
        p += sizeof(inner);
        const std::uint64_t count = words;
        std::memcpy(p, &count, 8);
        p += 8;
        if (words)
            SEALHIP_CHECK(hipMemcpyAsync(p, src, words * 8, hipMemcpyDeviceToHost, e.lane().stream));
        e.sync_and_check(); // the words become host-visible here: a failed launch must not be saved with S_OK
        std::uint64_t first_c1 = 0;
        if (ci.size == 2 && words)
            std::memcpy(&first_c1, p + static_cast<std::size_t>(k) * e.n * 8, 8); // (the stream is not 8-byte aligned)
        if (ci.size == 2 && words && first_c1 == kSeedMarker)
        {
            // a ciphertext whose c_1 starts with the seed marker would be written in the seeded form by the reference
            // (ciphertext.cpp:189-208); evaluated ciphertexts never carry it
            throw std::logic_error("ciphertext carries a seed marker: save it on the host");
        }
        return total;
    }
    // KSwitchKeys::save (kswitchkeys.cpp:43-85 inside Serialization::Save): outer header, parms_id, keys_dim1, then per slot
    // keys_dim2 and that many PublicKey = Ciphertext streams (size 2, key level, NTT form, scale 1.0), whose words are copied
    // straight from the resident K1 buffer. keys[i] == nullptr: an unused slot (keys_dim2 = 0), as GaloisKeys has.
    std::size_t wire_kswitch_save_size(const Engine &e, const KSwitchKey *const *keys, std::size_t n_slots)
    {
        std::size_t total = sizeof(Header) + 32 + 8;
        for (std::size_t i = 0; i < n_slots; i++)
            total += 8 + (keys[i] ? keys[i]->n_digits * wire_save_size(2, static_cast<std::uint32_t>(e.n_key), e.n) : 0);
        return total;
    }

    std::size_t wire_save_kswitch_keys(Engine &e, const KSwitchKey *const *keys, std::size_t n_slots, void *bytes,
                                       std::size_t capacity)
    {
        std::array<std::uint64_t, 4> pid{};
        {
            std::lock_guard<std::mutex> lock(e.mu);
            const auto it = e.parms_ids.find(e.n_key);
            if (it == e.parms_ids.end())
                throw std::logic_error("the key level's parms_id is not registered (sealhip_context_set_parms_id)");
            pid = it->second;
        }
        const std::size_t total = wire_kswitch_save_size(e, keys, n_slots);
        if (capacity < total)
            throw std::invalid_argument("destination buffer is too small");
        const std::size_t digit_words = static_cast<std::size_t>(2) * e.n_key * e.n;
        unsigned char *p = static_cast<unsigned char *>(bytes);
        const Header outer{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, total };
        std::memcpy(p, &outer, sizeof(outer));
        p += sizeof(outer);
        std::memcpy(p, pid.data(), 32);
        const std::uint64_t dim1 = n_slots;
        std::memcpy(p + 32, &dim1, 8);
        p += 40;
        const double one = 1.0;
        for (std::size_t i = 0; i < n_slots; i++)
        {
            const std::uint64_t dim2 = keys[i] ? keys[i]->n_digits : 0;
            std::memcpy(p, &dim2, 8);
            p += 8;
            for (std::uint64_t j = 0; j < dim2; j++)
            {
                if (keys[i]->words != dim2 * digit_words)
                    throw std::logic_error("kswitch_keys is not valid for encryption parameters");
                const std::size_t ct_total = wire_save_size(2, static_cast<std::uint32_t>(e.n_key), e.n);
                const Header h{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, ct_total };
                std::memcpy(p, &h, sizeof(h));
                p += sizeof(h);
                std::memcpy(p, pid.data(), 32);
                p += 32;
                *p++ = 1; // keys are kept in NTT form (keygenerator.cpp:347-352)
                const std::uint64_t size64 = 2, n64 = e.n, k64 = static_cast<std::uint64_t>(e.n_key);
                std::memcpy(p, &size64, 8);
                std::memcpy(p + 8, &n64, 8);
                std::memcpy(p + 16, &k64, 8);
                std::memcpy(p + 24, &one, 8);
                p += 32;
                const Header inner{ kMagic, kHeaderSize, kVersionMajor, kVersionMinor, 0, 0, sizeof(Header) + 8 + digit_words * 8 };
                std::memcpy(p, &inner, sizeof(inner));
                p += sizeof(inner);
                const std::uint64_t count = digit_words;
                std::memcpy(p, &count, 8);
                p += 8;
                SEALHIP_CHECK(hipMemcpyAsync(p, keys[i]->d_data + j * digit_words, digit_words * 8, hipMemcpyDeviceToHost,
                                             e.lane().stream));
                p += digit_words * 8;
            }
        }
        e.sync_and_check();
        return total;
    }

    // KSwitchKeys::load (kswitchkeys.cpp:87-150): outer header, parms_id, keys_dim1, then per index keys_dim2 and that many
    // PublicKey streams, each a complete Ciphertext stream (publickey.h:107-111). The digits of keys_[index] are
    // concatenated straight into one device buffer, which is the K1 layout (keygenerator.cpp:325-369) the key switch reads.
    // Returns the number of digits; 0 when the slot is empty (a GaloisKeys object holds only the generated elements).
    std::uint32_t wire_load_kswitch_key(Engine &e, const void *bytes, std::size_t len, std::uint32_t index, u64 **d_out,
                                        std::size_t *words_out, std::uint64_t *dim1_out)
    {
        const unsigned char *p = static_cast<const unsigned char *>(bytes);
        const Header outer = read_header(p, len);
        if (outer.size > len || outer.size < sizeof(Header) + 32 + 8)
            throw std::runtime_error("I/O error");
        const unsigned char *end = p + outer.size;
        p += sizeof(Header);
        std::uint64_t pid[4], dim1;
        std::memcpy(pid, p, 32);
        std::memcpy(&dim1, p + 32, 8);
        p += 40;
        *dim1_out = dim1;
        {
            std::lock_guard<std::mutex> lock(e.mu);
            const auto key_id = e.parms_ids.find(e.n_key);
            if (key_id == e.parms_ids.end() || std::memcmp(key_id->second.data(), pid, 32) != 0)
                throw std::logic_error("kswitch_keys is not valid for encryption parameters"); // is_metadata_valid_for, valcheck.cpp
        }
        if (index >= dim1)
            throw std::invalid_argument("key index out of range");
        const std::size_t digit_words = static_cast<std::size_t>(2) * e.n_key * e.n;
        const std::uint32_t max_digits = static_cast<std::uint32_t>((e.k_first + e.nsp - 1) / e.nsp);
        for (std::uint64_t i = 0; i <= index; i++)
        {
            if (static_cast<std::size_t>(end - p) < 8)
                throw std::runtime_error("I/O error");
            std::uint64_t dim2;
            std::memcpy(&dim2, p, 8);
            p += 8;
            if (dim2 > 64)
                throw std::logic_error("kswitch_keys is not valid for encryption parameters");
            // owns the device buffer until the key is handed over: any failure (also of the final synchronisation)
            // drains the stream and frees it
            struct DevGuard
            {
                u64 *dev = nullptr;
                hipStream_t stream;
                ~DevGuard()
                {
                    if (dev)
                    {
                        (void)hipStreamSynchronize(stream);
                        (void)hipFree(dev);
                    }
                }
            } guard{ nullptr, e.lane().stream };
            if (i == index)
            {
                if (dim2 == 0)
                    return 0;
                if (dim2 > max_digits)
                    throw std::logic_error("kswitch_keys is not valid for encryption parameters");
                SEALHIP_CHECK(hipMalloc(reinterpret_cast<void **>(&guard.dev), dim2 * digit_words * sizeof(u64)));
            }
            u64 *const dev = guard.dev;
            for (std::uint64_t j = 0; j < dim2; j++)
            {
                {
                    const Parsed ps = parse(p, static_cast<std::size_t>(end - p));
                    if (i == index)
                    {
                        if (std::memcmp(ps.info.parms_id, pid, 32) != 0 || ps.info.size != 2 ||
                            ps.info.coeff_modulus_size != static_cast<std::uint32_t>(e.n_key) ||
                            ps.info.poly_modulus_degree != e.n || !ps.info.is_ntt_form)
                            throw std::logic_error("kswitch_keys is not valid for encryption parameters");
                        if (ps.info.seeded)
                        {
                            // keys saved through Serializable<> carry c_1 as a seed (keygenerator.cpp:325-369 with save_seed)
                            if (ps.info.data_words != digit_words / 2)
                                throw std::logic_error("kswitch_keys is not valid for encryption parameters");
                            const std::vector<u64> c1 = wire_expand_seed(e, e.n_key, ps.words + (digit_words / 2) * 8);
                            SEALHIP_CHECK(hipMemcpyAsync(dev + j * digit_words, ps.words, (digit_words / 2) * sizeof(u64),
                                                         hipMemcpyHostToDevice, e.lane().stream));
                            SEALHIP_CHECK(hipMemcpyAsync(dev + j * digit_words + digit_words / 2, c1.data(),
                                                         (digit_words / 2) * sizeof(u64), hipMemcpyHostToDevice, e.lane().stream));
                            SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream)); // c1 is a local buffer
                        }
                        else
                        {
                            if (ps.info.data_words != digit_words)
                                throw std::logic_error("kswitch_keys is not valid for encryption parameters");
                            SEALHIP_CHECK(hipMemcpyAsync(dev + j * digit_words, ps.words, digit_words * sizeof(u64),
                                                         hipMemcpyHostToDevice, e.lane().stream));
                        }
                    }
                    p += ps.info.total_bytes;
                }
            }
            if (i == index)
            {
                SEALHIP_CHECK(hipStreamSynchronize(e.lane().stream));
                guard.dev = nullptr; // handed over
                *d_out = dev;
                *words_out = dim2 * digit_words;
                return static_cast<std::uint32_t>(dim2);
            }
        }
        return 0;
    }
} // namespace sealhip
