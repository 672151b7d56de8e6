#!/usr/bin/env python3
"""Generates tests/golden/prng_vectors.json with the REFERENCE's own BLAKE2 code: oracle/_ref/libblake2ref.so is compiled by
`make -C oracle ref` from /root/reference/native/src/seal/util/{blake2b,blake2xb}.c where they lie (nothing is copied).
Vectors: BLAKE2Xb outputs for a few (message, key, length) triples, and for three seeds the first and the 4096th byte
range of BlakePRNG buffers 0 and 1 (randomgen.cpp:63-73: buffer c = blake2xb(4096, in = c as 8 LE bytes, key = seed))."""
import ctypes as C
import json
import os
import struct

HERE = os.path.dirname(os.path.abspath(__file__))
lib = C.CDLL(os.path.join(HERE, "..", "..", "oracle", "_ref", "libblake2ref.so"))
lib.blake2xb.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t, C.c_void_p, C.c_size_t]
lib.blake2xb.restype = C.c_int


def xb(outlen, data, key=b""):
    out = C.create_string_buffer(outlen)
    assert lib.blake2xb(out, outlen, data, len(data), key if key else None, len(key)) == 0
    return out.raw


vec = {"blake2xb": [], "prng": []}
for outlen, data, key in ((1, b"", b""), (32, b"abc", b""), (64, b"abc", b"k" * 64), (65, bytes(range(200)), bytes(range(64))),
                          (200, bytes(range(129)), b""), (4096, struct.pack("<Q", 0), bytes(64))):
    vec["blake2xb"].append({"outlen": outlen, "data": data.hex(), "key": key.hex(), "digest_sha": None,
                            "head": xb(outlen, data, key)[:64].hex(), "tail": xb(outlen, data, key)[-16:].hex()})
for seed in ([0] * 8, list(range(1, 9)), [0x0123456789ABCDEF ^ (i * 0x9E3779B97F4A7C15 & 0xFFFFFFFFFFFFFFFF) for i in range(8)]):
    key = struct.pack("<8Q", *seed)
    bufs = [xb(4096, struct.pack("<Q", c), key) for c in (0, 1)]
    vec["prng"].append({"seed": [str(s) for s in seed], "buffer0_head": bufs[0][:32].hex(), "buffer0_tail": bufs[0][-32:].hex(),
                        "buffer1_head": bufs[1][:32].hex()})
for v in vec["blake2xb"]:
    del v["digest_sha"]
json.dump(vec, open(os.path.join(HERE, "prng_vectors.json"), "w"), indent=1)
print("wrote", len(vec["blake2xb"]), "+", len(vec["prng"]), "vectors")
