"""Crafted candidate-DENSE inputs for the parity tests (test infrastructure).

The Buzhash window hash of 64-byte-periodic data is itself periodic: every window holds each byte of the period exactly
once, and stepping one byte forward only rotates the hash by one bit (the incoming and the outgoing table value are the
same). So a 64-byte pattern whose window hash passes the break test at ONE phase gives one candidate per 64 bytes — and a
pattern whose window hash is 0xFFFFFFFF (rotation-invariant, passes `(h & mask) >= mask - 2` for every mask) makes EVERY
position a candidate. Periods that divide 32 cancel to h = 0 (each table value appears at two rotations 32 apart): no
candidate at all, max-size cuts only — like a zero run.

The reference's writer takes such bytes like any others (internal/pxarmount/commit_reuse.go:457,
internal/tapeio/converter.go:836: no content-dependent error); the serial chunker cuts them at `s + min` every time.
"""
import numpy as np


def _rotl(x, k):
    k &= 31
    x = np.asarray(x, dtype=np.uint64) & 0xFFFFFFFF
    return ((x << np.uint64(k)) | (x >> np.uint64((32 - k) & 31))) & np.uint64(0xFFFFFFFF) if k else x


def window_hash(table, window64) -> int:
    """h of one 64-byte window, oracle/buzhash_oracle.c's recurrence restated: h = XOR_k rotl(T[b[63 - k]], k mod 32)."""
    h = 0
    for b in window64:
        h = ((h << 1) | (h >> 31)) & 0xFFFFFFFF
        h ^= int(table[int(b)])
    return h


def all_candidate_pattern(table, seed: int = 1) -> np.ndarray:
    """A 64-byte pattern p with window_hash(p) == 0xFFFFFFFF: np.tile(p, n) has a candidate at EVERY position >= 64.
    Meet in the middle over the last four bytes (2 x 65 536 partial sums) behind a random 60-byte prefix."""
    T = np.asarray(table, dtype=np.uint64)
    rng = np.random.default_rng(seed)
    v = np.arange(256)
    for _ in range(64):
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        p[60:] = 0
        # contribution of byte j of the pattern when the window ends on byte 63: rotl(T[p_j], 63 - j)
        rest = 0
        for j in range(60):
            rest ^= int(_rotl(T[int(p[j])], 63 - j))
        target = rest ^ 0xFFFFFFFF
        left = (_rotl(T[v], 3)[:, None] ^ _rotl(T[v], 2)[None, :]).reshape(-1)      # bytes 60, 61
        right = (_rotl(T[v], 1)[:, None] ^ _rotl(T[v], 0)[None, :]).reshape(-1)     # bytes 62, 63
        order = np.argsort(left)
        ls = left[order]
        want = right ^ np.uint64(target)
        pos = np.searchsorted(ls, want)
        pos[pos >= ls.size] = 0
        hit = np.nonzero(ls[pos] == want)[0]
        if hit.size:
            r = int(hit[0])
            l = int(order[pos[r]])
            p[60], p[61], p[62], p[63] = l >> 8, l & 255, r >> 8, r & 255
            assert window_hash(table, p) == 0xFFFFFFFF
            return p
    raise AssertionError("no all-candidate pattern found")


def one_phase_pattern(O, cfg, seed: int = 5, tries: int = 40000):
    """A random 64-byte pattern whose periodic window hash passes the break test at (at least) one phase."""
    rng = np.random.default_rng(seed)
    for _ in range(tries):
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(p, 8)).size >= 6:
            return p
    return None


def crafted_stream(O, cfg, total: int, seed: int, allp: np.ndarray, onep, unit: int) -> np.ndarray:
    """`total` bytes made of stretches of random bytes, every-position-a-candidate bytes, one-candidate-per-64-bytes
    bytes, a period that cancels (no candidate at all) and zeros — stretch lengths from a few bytes to several `unit`s,
    boundaries at arbitrary (unaligned) offsets, so that dense tiles start and end anywhere inside scan tiles and pages."""
    rng = np.random.default_rng(seed)
    out = np.empty(total, dtype=np.uint8)
    pos = 0
    kinds = [0, 1, 2, 1, 0, 3, 1, 4, 2, 0]
    i = 0
    while pos < total:
        kind = kinds[i % len(kinds)]
        i += 1
        n = int(rng.integers(1, 4 * unit)) if rng.random() < 0.8 else int(rng.integers(1, 200))
        n = min(n, total - pos)
        if kind == 0:
            out[pos:pos + n] = O.fill(n, seed * 131 + i, 0)
        elif kind == 1:
            ph = int(rng.integers(0, 64))
            out[pos:pos + n] = np.tile(np.roll(allp, -ph), n // 64 + 2)[:n]
        elif kind == 2 and onep is not None:
            out[pos:pos + n] = np.tile(onep, n // 64 + 2)[:n]
        elif kind == 3:
            out[pos:pos + n] = np.tile(allp[:16], n // 16 + 2)[:n]      # period 16: the window hash cancels to 0
        else:
            out[pos:pos + n] = 0
        pos += n
    return out
