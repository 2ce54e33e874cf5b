"""Whole-batch parity check against the CPU oracle at RESTART POINTS (TEST INFRASTRUCTURE ONLY).

A cut is a restart point of the serial chunker (Proxmox ``ChunkerImpl::scan`` resets ``h``,
``window_size`` and ``chunk_size`` on every boundary, oracle/buzhash_oracle.c), so the records that
follow a cut depend only on the bytes behind it. That makes a 64 GiB batch checkable anywhere
without running the byte-serial oracle over 64 GiB: take a cut the GPU reported at stream offset c
(any offset — beyond 2^32, beyond 2^35, in the last scan tile, in any resident slot), download
``span`` bytes from c, run the oracle from FRESH state on them and require its cuts and digests to
be the GPU's next records, one by one. Every oracle record but the last is decided by bytes inside
the span (the last one is the forced cut at the end of the download) — unless the span reaches the
end of the segment, then the last record is the segment's true final chunk and must match as well.

Only ``tests/`` and ``bench.py``'s ``cpu_baseline`` leg import this module (it runs the oracle).
"""
from __future__ import annotations

import threading

import numpy as np

from . import oracle as O


def _segment_slices(recs, nseg):
    """records are ordered by (segment, end): [lo, hi) of every segment's records"""
    seg = recs["segment"].astype(np.int64)
    lo = np.searchsorted(seg, np.arange(nseg), side="left")
    hi = np.searchsorted(seg, np.arange(nseg), side="right")
    return lo, hi


def pick_points(recs, segs, nbytes, k, span):
    """k restart points spread uniformly over the batch's bytes, plus the first cut beyond 2^32 and 2^35, the last
    point whose span reaches the end of the last segment (the final chunk and the last scan tile) and the start of the
    first segment. A point = (segment index, record index whose END is the restart offset, or -1 = segment start)."""
    nseg = len(segs)
    lo, hi = _segment_slices(recs, nseg)
    seg_off = np.array([s[0] for s in segs], dtype=np.uint64)
    seg_len = np.array([s[1] for s in segs], dtype=np.uint64)
    absend = seg_off[recs["segment"]] + recs["end"]                 # global END offset of every record (ascending)
    pts = set()

    def at_or_before(t):
        j = int(np.searchsorted(absend, np.uint64(t), side="right")) - 1
        if j < 0:
            return (0, -1)
        s = int(recs["segment"][j])
        if int(recs["end"][j]) >= int(seg_len[s]):                 # the segment's final record: restart = next segment
            return (s + 1, -1) if s + 1 < nseg else (s, int(hi[s]) - 2 if hi[s] - lo[s] >= 2 else -1)
        return (s, j)

    for i in range(k):
        pts.add(at_or_before((i + 0.5) / k * nbytes))
    for t in (1 << 32, 1 << 35, nbytes - 1):
        if nbytes > t:
            pts.add(at_or_before(t + 1 if t != nbytes - 1 else t))
    # the tail: the latest cut of the last segment from which `span` still covers the segment end
    s = nseg - 1
    if hi[s] > lo[s]:
        ends = recs["end"][lo[s]:hi[s]].astype(np.int64)
        ok = np.flatnonzero(int(seg_len[s]) - ends <= span)
        ok = ok[ends[ok] < int(seg_len[s])]
        pts.add((s, int(lo[s] + ok[0])) if ok.size else (s, -1))
    pts.add((0, -1))
    out = []
    for s, j in sorted(pts):
        if 0 <= s < nseg and (j < 0 or recs["segment"][j] == s):
            out.append((s, j))
    return out, lo, hi


def check_batch(download, segs, recs, avg, nbytes=None, k=32, span=64 << 20, threads=16, impl=1):
    """Compare the GPU's records `recs` of one batch with the oracle at restart points.

    download(offset, n) -> uint8 array of the batch's device bytes (what the GPU actually chunked)
    segs                [(offset, length)] of the batch (None = one segment covering nbytes)
    Returns {"points", "records_checked", "bytes_checked", "ok", "max_offset", "mismatch"}."""
    if segs is None:
        segs = [(0, int(nbytes))]
    segs = [(int(a), int(b)) for a, b in segs]
    if nbytes is None:
        nbytes = max(a + b for a, b in segs)
    cfg = O.new_config(avg)
    span = max(int(span), 4 * int(cfg.max))
    pts, lo, hi = pick_points(recs, segs, int(nbytes), k, span)
    res = {"points": len(pts), "records_checked": 0, "bytes_checked": 0, "ok": True, "max_offset": 0, "mismatch": None}
    lock = threading.Lock()

    def one(pt):
        s, j = pt
        so, sl = segs[s]
        rel = 0 if j < 0 else int(recs["end"][j])
        n = min(span, sl - rel)
        if n <= 0:
            return
        host = download(so + rel, n)
        want = O.chunk_and_digest(cfg, host, [(0, n)], impl=impl)
        to_end = (rel + n == sl)
        m = want.size if to_end else want.size - 1                  # the forced cut at the end of a partial span is not a cut
        first = int(lo[s]) if j < 0 else j + 1
        got = recs[first:first + m]
        good = (got.size == m and bool(np.all(got["segment"] == s))
                and np.array_equal(got["end"].astype(np.int64) - rel, want["end"][:m].astype(np.int64))
                and np.array_equal(got["digest"], want["digest"][:m])
                and np.array_equal(got["size"], want["size"][:m]))
        if to_end and good:                                         # ... and nothing may follow the segment's final record
            good = (first + m == int(hi[s]))
        with lock:
            res["records_checked"] += int(m)
            res["bytes_checked"] += int(n)
            res["max_offset"] = max(res["max_offset"], so + rel + n)
            if not good and res["ok"]:
                res["ok"] = False
                res["mismatch"] = {"segment": s, "restart_offset": so + rel, "span": n, "oracle_records": int(want.size),
                                   "gpu_records_available": int(got.size)}

    pts = list(pts)
    nth = max(1, min(threads, len(pts)))
    it = iter(pts)
    itlock = threading.Lock()

    def worker():
        while True:
            with itlock:
                pt = next(it, None)
            if pt is None:
                return
            one(pt)

    ths = [threading.Thread(target=worker) for _ in range(nth)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return res
