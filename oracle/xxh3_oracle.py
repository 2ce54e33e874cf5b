"""Pure-Python restatement of XXH3-64 (seed 0, default secret) — TEST INFRASTRUCTURE ONLY.

The reference tees new file bodies through github.com/zeebo/xxh3 (`xxh3.New()` ... `Sum64()`,
internal/pxarmount/commit_reuse.go:450-461) and re-hashes them after the commit
(internal/pxarmount/commit_orchestrate.go:485-562). zeebo/xxh3 implements the standard XXH3-64;
this file states that algorithm so the HIP kernel has something readable to be compared with,
and tests/test_oracle_xxh3.py pins it against the independent `xxhash` C library.
"""
M64 = (1 << 64) - 1
P32_1, P32_2, P32_3 = 0x9E3779B1, 0x85EBCA77, 0xC2B2AE3D
P64_1, P64_2, P64_3 = 0x9E3779B185EBCA87, 0xC2B2AE3D27D4EB4F, 0x165667B19E3779F9
P64_4, P64_5 = 0x85EBCA77C2B2AE63, 0x27D4EB2F165667C5
PMX1, PMX2 = 0x165667919E3779F9, 0x9FB21C651E98DF25

SECRET = bytes.fromhex(
    "b8fe6c3923a44bbe7c01812cf721ad1cded46de9839097db7240a4a4b7b3671f"
    "cb79e64eccc0e578825ad07dccff7221b8084674f743248ee03590e6813a264c"
    "3c2852bb91c300cb88d0658b1b532ea371644897a20df94e3819ef46a9deacd8"
    "a8fa763fe39c343ff9dcbbc7c70b4f1d8a51e04bcdb45931c89f7ec9d9787364"
    "eac5ac8334d3ebc3c581a0fffa1363eb170ddd51b7f0da49d316552629d4689e"
    "2b16be587d47a1fc8ff8b8d17ad031ce45cb3a8f95160428afd7fbcabb4b407e")
assert len(SECRET) == 192


def r64(b, o):
    return int.from_bytes(b[o:o + 8], "little")


def r32(b, o):
    return int.from_bytes(b[o:o + 4], "little")


def rotl64(x, n):
    return ((x << n) | (x >> (64 - n))) & M64


def swap64(x):
    return int.from_bytes(x.to_bytes(8, "little"), "big")


def mul128_fold64(a, b):
    p = a * b
    return (p & M64) ^ (p >> 64)


def avalanche(h):
    h ^= h >> 37
    h = (h * PMX1) & M64
    return h ^ (h >> 32)


def xxh64_avalanche(h):
    h ^= h >> 33
    h = (h * P64_2) & M64
    h ^= h >> 29
    h = (h * P64_3) & M64
    return h ^ (h >> 32)


def rrmxmx(h, n):
    h ^= rotl64(h, 49) ^ rotl64(h, 24)
    h = (h * PMX2) & M64
    h ^= ((h >> 35) + n) & M64
    h = (h * PMX2) & M64
    return h ^ (h >> 28)


def mix16(d, o, so):
    return mul128_fold64(r64(d, o) ^ r64(SECRET, so), r64(d, o + 8) ^ r64(SECRET, so + 8))


def xxh3_64(d: bytes) -> int:
    n = len(d)
    if n == 0:
        return xxh64_avalanche(r64(SECRET, 56) ^ r64(SECRET, 64))
    if n <= 3:
        comb = (d[0] << 16) | (d[n >> 1] << 24) | d[n - 1] | (n << 8)
        return xxh64_avalanche(comb ^ ((r32(SECRET, 0) ^ r32(SECRET, 4)) & 0xFFFFFFFF))
    if n <= 8:
        in1, in2 = r32(d, 0), r32(d, n - 4)
        keyed = (in2 + (in1 << 32)) ^ (r64(SECRET, 8) ^ r64(SECRET, 16))
        return rrmxmx(keyed, n)
    if n <= 16:
        lo = r64(d, 0) ^ (r64(SECRET, 24) ^ r64(SECRET, 32))
        hi = r64(d, n - 8) ^ (r64(SECRET, 40) ^ r64(SECRET, 48))
        return avalanche((n + swap64(lo) + hi + mul128_fold64(lo, hi)) & M64)
    if n <= 128:
        acc = (n * P64_1) & M64
        if n > 32:
            if n > 64:
                if n > 96:
                    acc += mix16(d, 48, 96) + mix16(d, n - 64, 112)
                acc += mix16(d, 32, 64) + mix16(d, n - 48, 80)
            acc += mix16(d, 16, 32) + mix16(d, n - 32, 48)
        acc += mix16(d, 0, 0) + mix16(d, n - 16, 16)
        return avalanche(acc & M64)
    if n <= 240:
        acc = (n * P64_1) & M64
        for i in range(8):
            acc += mix16(d, 16 * i, 16 * i)
        acc = avalanche(acc & M64)
        for i in range(8, n // 16):
            acc += mix16(d, 16 * i, 16 * (i - 8) + 3)
        acc += mix16(d, n - 16, 136 - 17)
        return avalanche(acc & M64)
    # long input: 8 accumulators, 64-byte stripes, 1024-byte blocks
    acc = [P32_3, P64_1, P64_2, P64_3, P64_4, P32_2, P64_5, P32_1]

    def accumulate(off, soff):
        vals = [r64(d, off + 8 * i) for i in range(8)]
        for i in range(8):
            key = vals[i] ^ r64(SECRET, soff + 8 * i)
            acc[i ^ 1] = (acc[i ^ 1] + vals[i]) & M64
            acc[i] = (acc[i] + (key & 0xFFFFFFFF) * (key >> 32)) & M64

    nb_blocks = (n - 1) // 1024
    for b in range(nb_blocks):
        for s in range(16):
            accumulate(b * 1024 + s * 64, s * 8)
        for i in range(8):
            a = acc[i]
            a ^= a >> 47
            a ^= r64(SECRET, 128 + 8 * i)
            acc[i] = (a * P32_1) & M64
    nb_stripes = ((n - 1) - 1024 * nb_blocks) // 64
    for s in range(nb_stripes):
        accumulate(nb_blocks * 1024 + s * 64, s * 8)
    accumulate(n - 64, 192 - 64 - 7)
    res = (n * P64_1) & M64
    for i in range(4):
        res += mul128_fold64(acc[2 * i] ^ r64(SECRET, 11 + 16 * i), acc[2 * i + 1] ^ r64(SECRET, 11 + 16 * i + 8))
    return avalanche(res & M64)
