"""Test helpers: a numpy model of the engine's resolve rule and golden-fixture I/O."""
import json
import os

import numpy as np

GOLDEN_DIR = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def resolve_model(cands, seg_len, cmin, cmax):
    """Python statement of kernels.hip k_resolve: given ascending candidate END offsets of one
    stream, apply the min/max rules. Must equal the serial chunker on every input."""
    effmin = max(cmin, 65)
    cands = np.asarray(cands, dtype=np.uint64)
    ends = []
    s = 0
    while s < seg_len:
        tlo, thi = s + effmin, s + cmax
        j = int(np.searchsorted(cands, tlo, side="left"))
        c = int(cands[j]) if j < cands.size else None
        e = c if (c is not None and c < thi) else thi
        e = min(e, seg_len)
        ends.append(e)
        s = e
    return np.asarray(ends, dtype=np.uint64)


def resolve_model_suggested(cands, sugg, seg_len, cmin, cmax):
    """k_resolve with suggested boundaries: after a cut at s the next cut is the EARLIER of the hash/max cut and the
    first suggested boundary b with b - s >= min (those closer than min are dropped for good); b wins only when
    b < hash cut <= s + max. Must equal the payload chunker fed byte by byte."""
    effmin = max(cmin, 65)
    cands = np.asarray(cands, dtype=np.uint64)
    sugg = np.asarray(sorted(sugg), dtype=np.uint64)
    ends = []
    s = 0
    while s < seg_len:
        tlo, thi = s + effmin, s + cmax
        j = int(np.searchsorted(cands, tlo, side="left"))
        c = int(cands[j]) if j < cands.size else None
        e = c if (c is not None and c < thi) else thi
        e = min(e, seg_len)
        k = int(np.searchsorted(sugg, s + cmin, side="left"))
        if k < sugg.size and int(sugg[k]) < e:
            e = int(sugg[k])
        ends.append(e)
        s = e
    return np.asarray(ends, dtype=np.uint64)


def records_equal(a, b):
    if a.shape != b.shape:
        return False
    return (np.array_equal(a["end"], b["end"]) and np.array_equal(a["segment"], b["segment"])
            and np.array_equal(a["size"], b["size"]) and np.array_equal(a["digest"], b["digest"]))


def describe_mismatch(got, want, limit=5):
    lines = [f"got {got.size} records, want {want.size}"]
    n = min(got.size, want.size)
    bad = [i for i in range(n) if got["end"][i] != want["end"][i] or got["segment"][i] != want["segment"][i]
           or not np.array_equal(got["digest"][i], want["digest"][i])][:limit]
    for i in bad:
        lines.append(f"  [{i}] got seg={got['segment'][i]} end={got['end'][i]} size={got['size'][i]} "
                     f"dig={bytes(got['digest'][i]).hex()[:16]} | want seg={want['segment'][i]} end={want['end'][i]} "
                     f"size={want['size'][i]} dig={bytes(want['digest'][i]).hex()[:16]}")
    return "\n".join(lines)


def load_golden(name):
    with open(os.path.join(GOLDEN_DIR, name)) as f:
        return json.load(f)


def golden_files():
    """every tests/golden/chunks_*.json: the committed fixture (the oracle's own output) and — once a maintainer has run
    `make golden-go` — chunks_go.json from the REAL Go module (tools/golden/README.md), picked up automatically"""
    return sorted(f for f in os.listdir(GOLDEN_DIR) if f.startswith("chunks_") and f.endswith(".json"))


def golden_case_data(O, case):
    """(bytes, segment table) of one golden case, rebuilt from its generator description"""
    if case["segments"] == "le_u32_counter_262144":
        data = np.arange(256 * 1024, dtype="<u4").view(np.uint8)
        return data, [(0, data.size)]
    parts, table, off = [], [], 0
    for s in case["segments"]:
        if "pattern" in s:   # a crafted period (round 6: candidate-dense data): the pattern's bytes repeated
            pat = np.frombuffer(bytes.fromhex(s["pattern"]), dtype=np.uint8)
            parts.append(np.tile(pat, s["length"] // pat.size + 1)[: s["length"]].copy())
            table.append((off, s["length"]))
            off += s["length"]
            continue
        parts.append(O.fill(s["length"], s["seed"], s["kind"]))
        table.append((off, s["length"]))
        off += s["length"]
    return np.concatenate(parts), table


def golden_suggested_data(O, case):
    """bytes of a `suggested` case of schema v2 (tools/golden/main.go): the LE-u32 counter buffer or a splitmix64 stream"""
    n = int(case["length"])
    if case["name"].startswith("counter"):
        return np.arange(n // 4, dtype="<u4").view(np.uint8)
    return O.fill(n, int(case["seed"]), 0)


def golden_records(case, dtype):
    out = np.zeros(len(case["records"]), dtype=dtype)
    for i, (seg, end, size, dig) in enumerate(case["records"]):
        out[i]["segment"], out[i]["end"], out[i]["size"] = seg, end, size
        out[i]["digest"] = np.frombuffer(bytes.fromhex(dig), dtype=np.uint8)
    return out


def scan_lane_model(data, table, mask, break_min, strip=4352):
    """numpy statement of kernels.hip k_scan2's per-lane algebra, for one buffer cut into lane strips:
      * prefix recurrence P(i) = rotl(P(i-1), 1) ^ T'[b[i]] restarted 64 bytes before every strip,
      * window hash h'(i) = P(i) ^ P(i-64)   (64 = 0 mod 32, so the leaving byte needs no rotation),
      * table pre-rotated by r = 32 - bits, candidate test h' >= break_min << r (one unsigned compare).
    Returns ascending candidate END offsets; must equal oracle.candidates()."""
    data = np.asarray(data, dtype=np.uint8)
    bits = int(mask + 1).bit_length() - 1
    assert (1 << bits) == mask + 1
    r = (32 - bits) & 31
    T = np.asarray(table, dtype=np.uint64)
    Trot = (((T << r) | (T >> (32 - r))) & 0xFFFFFFFF) if r else T.copy()
    thr = (int(break_min) << r) & 0xFFFFFFFF
    n = data.size
    out = []
    for s0 in range(0, n, strip):
        lo = max(0, s0 - 64)
        P = 0
        ring = [0] * 64
        # warm-up over the 64 bytes before the strip (zeros before the buffer start)
        warm = [0] * (64 - (s0 - lo)) + data[lo:s0].tolist()
        for k, b in enumerate(warm):
            P = (((P << 1) | (P >> 31)) & 0xFFFFFFFF) ^ int(Trot[b])
            ring[k] = P
        for i in range(s0, min(n, s0 + strip)):
            rp = ((P << 1) | (P >> 31)) & 0xFFFFFFFF
            tv = int(Trot[data[i]])
            ri = (i - s0) & 63
            h = rp ^ tv ^ ring[ri]
            P = rp ^ tv
            ring[ri] = P
            e = i + 1
            if h >= thr and e >= 64:
                out.append(e)
    return np.asarray(out, dtype=np.uint64)


def scan_coop_model(data, table, mask, break_min, lines=4):
    """numpy/python statement of kernels.hip k_scan3's dataflow for one buffer (no GPU):
      * wave tiles of 64 strips x (lines*128) bytes, a strip per lane, 64-byte half-lines;
      * quad-cooperative fetch: load j of a half-line gives lane (q, ql) piece ql (16 B) of strip 4q+j, zero outside
        the buffer; the per-wave stage [strip][piece] hands every lane its own 64 bytes back (the LDS transpose);
      * rolling hash h(i) = rotl(h(i-1),1) ^ t(i) ^ t(i-64) with TWO 64-entry rings of table values that swap roles
        every half-line (warm-up plays half-line -1 and fills ring[1]; half-line hl writes ring[hl & 1]);
      * pre-rotated table, one unsigned compare.
    Returns ascending candidate END offsets; must equal oracle.candidates()."""
    data = np.asarray(data, dtype=np.uint8)
    bits = int(mask + 1).bit_length() - 1
    r = (32 - bits) & 31
    T = np.asarray(table, dtype=np.uint64)
    Trot = ((((T << r) | (T >> (32 - r))) & 0xFFFFFFFF) if r else T.copy()).astype(np.uint64)
    thr = (int(break_min) << r) & 0xFFFFFFFF
    n = data.size
    SL = lines * 128
    TILE = 64 * SL
    out = []

    def piece(a):  # 16 bytes at stream offset a, zeros outside [0, n)
        if a >= n or a + 16 <= 0:
            return np.zeros(16, dtype=np.uint8)
        v = np.zeros(16, dtype=np.uint8)
        lo, hi = max(a, 0), min(a + 16, n)
        v[lo - a:hi - a] = data[lo:hi]
        return v

    for wbase in range(0, n, TILE):
        ring = np.zeros((64, 2, 64), dtype=np.uint64)   # [lane][which][pos]
        h = np.zeros(64, dtype=np.uint64)
        for hl in range(-1, 2 * lines):
            stage = np.zeros((64, 4, 16), dtype=np.uint8)     # [strip][piece][byte]
            for lane in range(64):                            # the 4 loads of every lane
                q4, ql = lane & ~3, lane & 3
                for j in range(4):
                    stage[q4 + j, ql] = piece(wbase + (q4 + j) * SL + hl * 64 + ql * 16)
            rows = stage.reshape(64, 64)                      # lane's own row after the transpose
            new, old = (hl & 1), (hl & 1) ^ 1
            for pos in range(64):
                t = Trot[rows[:, pos]]
                hrot = ((h << np.uint64(1)) | (h >> np.uint64(31))) & np.uint64(0xFFFFFFFF)
                if hl < 0:
                    h = hrot ^ t                              # warm-up: no leaving term yet
                else:
                    h = hrot ^ t ^ ring[:, old, pos]
                    ends = wbase + np.arange(64) * SL + hl * 64 + pos + 1
                    hit = (h >= thr) & (ends >= 64) & (ends <= n)
                    out.extend(int(e) for e in ends[hit])
                ring[:, new, pos] = t
    return np.asarray(sorted(out), dtype=np.uint64)
