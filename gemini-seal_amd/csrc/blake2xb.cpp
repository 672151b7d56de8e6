// blake2xb.cpp -- see blake2xb.hpp. BLAKE2b follows RFC 7693 section 3 (G, F, the message schedule SIGMA, the
// initialisation vector) with the full 64-byte parameter block of the BLAKE2 specification (fanout, depth, leaf length,
// node offset, XOF length, node depth, inner length); BLAKE2Xb follows section 2 of the BLAKE2X specification: a root hash
// H0 = BLAKE2b(64 bytes, XOF length = l) and output block i = BLAKE2b(H0; digest length min(64, rest), fanout 0, depth 0,
// leaf length 64, node offset i, XOF length l, inner length 64).
#include "blake2xb.hpp"

#include <cstring>

namespace sealhip
{
    namespace
    {
        constexpr std::uint64_t kIV[8] = { 0x6a09e667f3bcc908ULL, 0xbb67ae8584caa73bULL, 0x3c6ef372fe94f82bULL,
                                           0xa54ff53a5f1d36f1ULL, 0x510e527fade682d1ULL, 0x9b05688c2b3e6c1fULL,
                                           0x1f83d9abfb41bd6bULL, 0x5be0cd19137e2179ULL };
        constexpr unsigned char kSigma[12][16] = {
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 },
            { 11, 8, 12, 0, 5, 2, 15, 13, 10, 14, 3, 6, 7, 1, 9, 4 }, { 7, 9, 3, 1, 13, 12, 11, 14, 2, 6, 5, 10, 4, 0, 15, 8 },
            { 9, 0, 5, 7, 2, 4, 10, 15, 14, 1, 11, 12, 6, 8, 3, 13 }, { 2, 12, 6, 10, 0, 11, 8, 3, 4, 13, 7, 5, 15, 14, 1, 9 },
            { 12, 5, 1, 15, 14, 13, 4, 10, 0, 7, 6, 3, 9, 2, 8, 11 }, { 13, 11, 7, 14, 12, 1, 3, 9, 5, 0, 15, 4, 8, 6, 2, 10 },
            { 6, 15, 14, 9, 11, 3, 0, 8, 12, 2, 13, 7, 1, 4, 10, 5 }, { 10, 2, 8, 4, 7, 6, 1, 5, 15, 11, 9, 14, 3, 12, 13, 0 },
            { 0, 1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 12, 13, 14, 15 }, { 14, 10, 4, 8, 9, 15, 13, 6, 1, 12, 0, 2, 11, 7, 5, 3 }
        };

        inline std::uint64_t rotr(std::uint64_t x, int r)
        {
            return (x >> r) | (x << (64 - r));
        }
        inline std::uint64_t load64(const unsigned char *p)
        {
            std::uint64_t v = 0;
            for (int i = 7; i >= 0; i--)
                v = (v << 8) | p[i];
            return v;
        }

        struct B2
        {
            std::uint64_t h[8], t0 = 0, t1 = 0;
            unsigned char buf[128];
            std::size_t buflen = 0, outlen = 64;

            // the 64-byte parameter block, XORed into the IV (BLAKE2 specification, section 2.5 / 2.8)
            void init(unsigned digest_length, unsigned key_length, unsigned fanout, unsigned depth, std::uint32_t leaf_length,
                      std::uint32_t node_offset, std::uint32_t xof_length, unsigned node_depth, unsigned inner_length)
            {
                unsigned char p[64] = { 0 };
                p[0] = static_cast<unsigned char>(digest_length);
                p[1] = static_cast<unsigned char>(key_length);
                p[2] = static_cast<unsigned char>(fanout);
                p[3] = static_cast<unsigned char>(depth);
                for (int i = 0; i < 4; i++)
                {
                    p[4 + i] = static_cast<unsigned char>(leaf_length >> (8 * i));
                    p[8 + i] = static_cast<unsigned char>(node_offset >> (8 * i));
                    p[12 + i] = static_cast<unsigned char>(xof_length >> (8 * i));
                }
                p[16] = static_cast<unsigned char>(node_depth);
                p[17] = static_cast<unsigned char>(inner_length);
                for (int i = 0; i < 8; i++)
                    h[i] = kIV[i] ^ load64(p + 8 * i);
                outlen = digest_length;
                t0 = t1 = 0;
                buflen = 0;
            }
            void compress(const unsigned char *block, bool last)
            {
                std::uint64_t m[16], v[16];
                for (int i = 0; i < 16; i++)
                    m[i] = load64(block + 8 * i);
                for (int i = 0; i < 8; i++)
                {
                    v[i] = h[i];
                    v[8 + i] = kIV[i];
                }
                v[12] ^= t0;
                v[13] ^= t1;
                if (last)
                    v[14] = ~v[14];
                auto G = [&](int r, int i, int a, int b, int c, int d) {
                    v[a] = v[a] + v[b] + m[kSigma[r][2 * i]];
                    v[d] = rotr(v[d] ^ v[a], 32);
                    v[c] = v[c] + v[d];
                    v[b] = rotr(v[b] ^ v[c], 24);
                    v[a] = v[a] + v[b] + m[kSigma[r][2 * i + 1]];
                    v[d] = rotr(v[d] ^ v[a], 16);
                    v[c] = v[c] + v[d];
                    v[b] = rotr(v[b] ^ v[c], 63);
                };
                for (int r = 0; r < 12; r++)
                {
                    G(r, 0, 0, 4, 8, 12);
                    G(r, 1, 1, 5, 9, 13);
                    G(r, 2, 2, 6, 10, 14);
                    G(r, 3, 3, 7, 11, 15);
                    G(r, 4, 0, 5, 10, 15);
                    G(r, 5, 1, 6, 11, 12);
                    G(r, 6, 2, 7, 8, 13);
                    G(r, 7, 3, 4, 9, 14);
                }
                for (int i = 0; i < 8; i++)
                    h[i] ^= v[i] ^ v[8 + i];
            }
            void update(const unsigned char *in, std::size_t inlen)
            {
                while (inlen)
                {
                    if (buflen == 128) // the buffer is only compressed when more input follows: the last block is special
                    {
                        t0 += 128;
                        t1 += t0 < 128;
                        compress(buf, false);
                        buflen = 0;
                    }
                    const std::size_t take = inlen < 128 - buflen ? inlen : 128 - buflen;
                    std::memcpy(buf + buflen, in, take);
                    buflen += take;
                    in += take;
                    inlen -= take;
                }
            }
            void final(unsigned char *out)
            {
                t0 += buflen;
                t1 += t0 < buflen;
                std::memset(buf + buflen, 0, 128 - buflen);
                compress(buf, true);
                for (std::size_t i = 0; i < outlen; i++)
                    out[i] = static_cast<unsigned char>(h[i >> 3] >> (8 * (i & 7)));
            }
        };
    } // namespace

    bool blake2xb(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key, std::size_t keylen)
    {
        if (!out || outlen == 0 || outlen > 0xFFFFFFFFull || keylen > 64 || (!key && keylen) || (!in && inlen))
            return false;
        B2 root;
        root.init(64, static_cast<unsigned>(keylen), 1, 1, 0, 0, static_cast<std::uint32_t>(outlen), 0, 0);
        if (keylen) // a keyed hash starts with the key padded to a full block
        {
            unsigned char block[128] = { 0 };
            std::memcpy(block, key, keylen);
            root.update(block, 128);
        }
        root.update(static_cast<const unsigned char *>(in), inlen);
        unsigned char h0[64];
        root.final(h0);
        unsigned char *o = static_cast<unsigned char *>(out);
        std::size_t rest = outlen;
        for (std::uint32_t i = 0; rest; i++)
        {
            const std::size_t take = rest < 64 ? rest : 64;
            B2 node;
            node.init(static_cast<unsigned>(take), 0, 0, 0, 64, i, static_cast<std::uint32_t>(outlen), 0, 64);
            node.update(h0, 64);
            node.final(o);
            o += take;
            rest -= take;
        }
        return true;
    }

    BlakePrng::BlakePrng(const std::uint64_t (&s)[8])
    {
        std::memcpy(seed, s, sizeof(seed));
    }

    std::uint32_t BlakePrng::generate()
    {
        if (head == sizeof(buffer)) // randomgen.cpp:63-73 (the counter is hashed as its 8 little-endian bytes)
        {
            unsigned char ctr[8];
            for (int i = 0; i < 8; i++)
                ctr[i] = static_cast<unsigned char>(counter >> (8 * i));
            unsigned char key[64];
            for (int i = 0; i < 8; i++)
                for (int b = 0; b < 8; b++)
                    key[8 * i + b] = static_cast<unsigned char>(seed[i] >> (8 * b));
            (void)blake2xb(buffer, sizeof(buffer), ctr, sizeof(ctr), key, sizeof(key));
            counter++;
            head = 0;
        }
        std::uint32_t v = 0;
        for (int i = 3; i >= 0; i--)
            v = (v << 8) | buffer[head + static_cast<std::size_t>(i)];
        head += 4;
        return v;
    }

    void sample_poly_uniform(BlakePrng &prng, const std::uint64_t *moduli, std::size_t rows, std::size_t n, std::uint64_t *dst)
    {
        constexpr std::uint64_t max_random = 0x7FFFFFFFFFFFFFFFULL; // rlwe.cpp:113
        for (std::size_t j = 0; j < rows; j++)
        {
            const std::uint64_t q = moduli[j];
            const std::uint64_t max_multiple = max_random - (max_random % q) - 1; // :117 (barrett_reduce_63 is exact)
            for (std::size_t i = 0; i < n; i++)
            {
                std::uint64_t r;
                do
                {
                    const std::uint64_t hi = prng.generate(), lo = prng.generate(); // :124, in this order
                    r = (hi << 31) | (lo >> 1);
                } while (r >= max_multiple);
                dst[j * n + i] = r % q;
            }
        }
    }
} // namespace sealhip
