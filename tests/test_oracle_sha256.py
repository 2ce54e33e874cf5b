"""Pins the oracle's SHA-256 (oracle/sha256_oracle.c) against FIPS 180-4 known answers and
Python hashlib (OpenSSL). The chunk digest the reference consumes (datastore.ChunkInfo.Digest,
internal/pxarmount/commit_reuse.go:105-115) and verification.HashFile
(internal/agent/verification/handler.go:36-68) are both plain SHA-256."""
import ctypes as C
import hashlib

import numpy as np
import pytest

NIST = [
    (b"", "e3b0c44298fc1c149afbf4c8996fb92427ae41e4649b934ca495991b7852b855"),
    (b"abc", "ba7816bf8f01cfea414140de5dae2223b00361a396177a9cb410ff61f20015ad"),
    (b"abcdbcdecdefdefgefghfghighijhijkijkljklmklmnlmnomnopnopq",
     "248d6a61d20638b8e5c026930c3e6039a33ce45964ff2167f6ecedd419db06c1"),
    (b"abcdefghbcdefghicdefghijdefghijkefghijklfghijklmghijklmnhijklmnoijklmnopjklmnopqklmnopqrlmnopqrsmnopqrstnopqrstu",
     "cf5b16a778af8380036ce59e7b0492370b249b11e8f07a51afac45037afee9d1"),
    (b"a" * 1_000_000, "cdc76e5c9914fb9281a1c7e284d73e67f1809a48a497200e046d39ccc7112cd0"),
]


@pytest.mark.parametrize("impl", [0, 1])
def test_nist_vectors(O, impl):
    for msg, want in NIST:
        assert O.sha256(msg, impl).hex() == want


@pytest.mark.parametrize("impl", [0, 1])
def test_against_hashlib_all_padding_lengths(O, impl):
    rng = np.random.default_rng(5)
    blob = rng.integers(0, 256, 4096, dtype=np.uint8)
    for n in list(range(0, 260)) + [511, 512, 513, 1000, 4095, 4096]:
        assert O.sha256(blob[:n], impl) == hashlib.sha256(blob[:n].tobytes()).digest(), n


def test_scalar_and_shani_agree_on_large_input(O):
    d = O.fill((3 << 20) + 5, 99)
    assert O.sha256(d, 0) == O.sha256(d, 1) == hashlib.sha256(d.tobytes()).digest()


def test_incremental_update_any_split(O):
    L = O.lib()
    from oracle.oracle import _buf

    class Ctx(C.Structure):
        _fields_ = [("state", C.c_uint32 * 8), ("nbytes", C.c_uint64), ("buf", C.c_uint8 * 64),
                    ("buflen", C.c_uint32), ("impl", C.c_int)]

    L.oracle_sha256_init.argtypes = [C.POINTER(Ctx), C.c_int]
    L.oracle_sha256_update.argtypes = [C.POINTER(Ctx), C.c_void_p, C.c_size_t]
    L.oracle_sha256_final.argtypes = [C.POINTER(Ctx), C.c_void_p]
    rng = np.random.default_rng(6)
    data = rng.integers(0, 256, 10_000, dtype=np.uint8)
    want = hashlib.sha256(data.tobytes()).digest()
    for impl in (0, 1):
        for _ in range(20):
            cuts = np.sort(rng.integers(0, data.size, 7))
            ctx = Ctx()
            L.oracle_sha256_init(C.byref(ctx), impl)
            prev = 0
            for c in list(cuts) + [data.size]:
                piece = _buf(data[prev:c])
                L.oracle_sha256_update(C.byref(ctx), piece.ctypes.data if piece.size else None, piece.size)
                prev = c
            out = (C.c_uint8 * 32)()
            L.oracle_sha256_final(C.byref(ctx), out)
            assert bytes(out) == want
