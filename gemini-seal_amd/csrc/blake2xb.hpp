// blake2xb.hpp -- BLAKE2b (RFC 7693) and the BLAKE2Xb extendable-output function, written from the specifications, and the
// reference's seeded uniform sampler on top of them. Host code: this is what Ciphertext::expand_seed needs
// (native/src/seal/ciphertext.cpp:126-133 -> BlakePRNG, randomgen.cpp:63-73 -> sample_poly_uniform, util/rlwe.cpp:101-129),
// so that seeded ciphertexts and keys can be ingested from the wire without the host library.
#pragma once

#include <cstddef>
#include <cstdint>

namespace sealhip
{
    // out[0..outlen) = BLAKE2Xb(in, key); outlen in 1 .. 2^32-1, keylen <= 64. Returns false on bad arguments.
    bool blake2xb(void *out, std::size_t outlen, const void *in, std::size_t inlen, const void *key, std::size_t keylen);

    // BlakePRNG (randomgen.h:199-222): 4096-byte buffers, buffer number `counter` = BLAKE2Xb(in = counter as 8 little-endian
    // bytes, key = the 64-byte seed); generate() hands out consecutive 32-bit words (randomgen.h:73-92).
    struct BlakePrng
    {
        std::uint64_t seed[8];
        std::uint64_t counter = 0;
        unsigned char buffer[4096];
        std::size_t head = sizeof(buffer);
        explicit BlakePrng(const std::uint64_t (&s)[8]);
        std::uint32_t generate();
    };

    // sample_poly_uniform (util/rlwe.cpp:101-129): rows x n words, row j uniform modulo moduli[j] by rejection from 63 bits
    void sample_poly_uniform(BlakePrng &prng, const std::uint64_t *moduli, std::size_t rows, std::size_t n, std::uint64_t *dst);
} // namespace sealhip
