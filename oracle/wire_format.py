"""oracle/wire_format.py -- CPU restatement of the reference's ciphertext wire format (TEST INFRASTRUCTURE ONLY).

Follows, field by field:
  Serialization::SEALHeader          native/src/seal/serialization.h:69-90 (magic 0xA15E :51, header size 0x10 :56)
  Serialization::LoadHeader          native/src/seal/serialization.cpp:137-176 (SEAL 3.4 header upgrade, serialization.h:309-320)
  Ciphertext::save_members / load    native/src/seal/ciphertext.cpp:170-226 / :228-330
  IntArray<T>::save_members          native/src/seal/intarray.h:592-620 (written through Serialization::Save, i.e. behind
                                     its own SEALHeader)
Pinning: the header known-answers of native/tests/seal/serialization.cpp:48-130 (sizeof == 16, field values, the 3.4
upgrade). The reference holds no byte fixture of a whole ciphertext (its tests only round-trip), so the member layout
is a restatement, not a captured vector.
"""
import struct

import numpy as np

MAGIC, HEADER_SIZE, VERSION = 0xA15E, 0x10, (3, 5)
HEADER = struct.Struct("<HBBBBHQ")
SEED_MARKER = 0xFFFFFFFFFFFFFFFF


def header(total, compr_mode=0, version=VERSION):
    return HEADER.pack(MAGIC, HEADER_SIZE, version[0], version[1], compr_mode, 0, total)


def header_3_4(total, compr_mode=0):
    """legacy_headers::SEALHeader_3_4 (serialization.h:309-320)"""
    return struct.pack("<HBBIQ", MAGIC, 0, compr_mode, total, 0)


def is_valid_header(raw):
    magic, hsize, major, minor, compr, _reserved, _size = HEADER.unpack(raw[:16])
    return magic == MAGIC and hsize == HEADER_SIZE and (major, minor) == VERSION and compr == 0


def load_header(raw, try_upgrade=True):
    fields = list(HEADER.unpack(raw[:16]))
    if try_upgrade and not is_valid_header(raw):
        magic, _zero, compr, size32, _res = struct.unpack("<HBBIQ", raw[:16])
        up = header(size32, compr)
        if is_valid_header(up):
            fields = list(HEADER.unpack(up))
    return dict(zip(("magic", "header_size", "version_major", "version_minor", "compr_mode", "reserved", "size"), fields))


def save_ciphertext(parms_id, is_ntt_form, size, n, k, scale, words, seed=None):
    """words: the uint64 coefficient words (size*k*n of them, or k*n when `seed` -- 64 bytes -- follows)"""
    body = struct.pack("<4Q", *parms_id) + struct.pack("<B", 1 if is_ntt_form else 0)
    body += struct.pack("<QQQd", size, n, k, scale)
    data = struct.pack("<Q", len(words)) + np.ascontiguousarray(words, dtype="<u8").tobytes()
    body += header(16 + len(data)) + data
    if seed is not None:
        assert len(seed) == 64
        body += seed
    return header(16 + len(body)) + body


def load_ciphertext(raw):
    h = load_header(raw)
    assert is_valid_header(header(h["size"], h["compr_mode"], (h["version_major"], h["version_minor"])))
    raw = raw[: h["size"]]
    p = 16
    parms_id = struct.unpack("<4Q", raw[p:p + 32])
    is_ntt = raw[p + 32] != 0
    size, n, k, scale = struct.unpack("<QQQd", raw[p + 33:p + 65])
    p += 65
    ih = load_header(raw[p:p + 16])
    count = struct.unpack("<Q", raw[p + 16:p + 24])[0]
    assert ih["size"] == 24 + 8 * count
    words = np.frombuffer(raw[p + 24:p + 24 + 8 * count], dtype="<u8")
    rest = raw[p + 24 + 8 * count:]
    return dict(parms_id=parms_id, is_ntt_form=is_ntt, size=size, n=n, k=k, scale=scale, words=words,
                seed=bytes(rest) if rest else None)


def save_kswitch_keys(parms_id, keys, n, n_key):
    """KSwitchKeys::save_members (kswitchkeys.cpp:43-85): keys[index] = list of digits, each a (2, n_key, n) uint64 array
    (a PublicKey = size-2 NTT-form ciphertext at the key level, publickey.h:107-111); an empty list = unused slot"""
    body = struct.pack("<4Q", *parms_id) + struct.pack("<Q", len(keys))
    for digits in keys:
        body += struct.pack("<Q", len(digits))
        for d in digits:
            body += save_ciphertext(parms_id, True, 2, n, n_key, 1.0, np.ascontiguousarray(d, dtype="<u8").reshape(-1))
    return header(16 + len(body)) + body
