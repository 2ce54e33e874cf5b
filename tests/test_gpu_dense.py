"""Candidate-DENSE data through every path, bit-exact against the CPU oracle (round 6).

The reference's writer seam never fails because of byte CONTENT (transfer.ArchiveWriter.WriteEntryReader:
internal/pxarmount/commit_reuse.go:457, commit_walk.go:475, internal/tapeio/converter.go:836); the serial chunker cuts a
stream whose every position is a candidate at `s + min` and goes on. Rounds 3-5 of this engine failed such a stream on
the page ring / stream writer (PBSGPU_E_DENSITY) once a scan tile held more candidates than it had slots. Since round 6
a tile that overflows is resolved exactly by on-demand re-scans inside the resolve walk (DenseTiles, kernels.h) — on the
ring, through pbsgpu_stream_* and in the batch path — and these tests pin that: crafted periods (tests/dense_inputs.py:
every position a candidate; one candidate per 64 bytes; periods that cancel to none) alone, mixed with ordinary bytes at
unaligned offsets, next to ordinary streams, at avg 4096 (the reference's test configuration, commit_walk_test.go:25),
64 Ki and 4 Mi (production, commit_orchestrate.go:144).
"""
import os
import sys
import time

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import dense_inputs as D  # noqa: E402
from helpers import describe_mismatch, records_equal  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(autouse=True)
def _short_idle_timeout(monkeypatch):
    monkeypatch.setenv("PBSGPU_RING_IDLE_TIMEOUT_S", "20")


@pytest.fixture(scope="module")
def allp(O):
    return D.all_candidate_pattern(O.default_table())


def _engine(avg, inflight=1):
    from pbs_plus_amd import Engine, buzhash

    return Engine(buzhash.NewConfig(avg), device=0, inflight=inflight)


def _same(got, want, what):
    assert got.size == want.size, (what, got.size, want.size, got["end"][:6], want["end"][:6])
    assert np.array_equal(got["end"], want["end"]), what
    assert np.array_equal(got["size"], want["size"]), what
    assert np.array_equal(got["digest"], want["digest"]), what


def _feed_host_streams(eng, ring, datas, timeout_s=180.0):
    """every stream's bytes page by page (H2D into reserved pages), all streams at once; records per stream"""
    L = eng._L
    page = ring.page_bytes
    sids = [ring.open() for _ in datas]
    offs = [0] * len(datas)
    got = [[] for _ in datas]
    done = [False] * len(datas)
    t0 = time.time()
    while not all(done):
        assert time.time() - t0 < timeout_s, ring.debug()
        for i, (sid, d) in enumerate(zip(sids, datas)):
            if done[i]:
                continue
            if offs[i] < d.size or (d.size == 0 and offs[i] == 0):
                if d.size == 0:
                    ring.commit(sid, 0, final=True)
                    offs[i] = 1
                else:
                    r = ring.reserve(sid)
                    if r is not None:
                        n = min(page, d.size - offs[i])
                        assert L.pbsgpu_memcpy_h2d(eng._h, r[0], d[offs[i]:offs[i] + n].ctypes.data, n) == 0
                        offs[i] += n
                        ring.commit(sid, n, final=(offs[i] == d.size))
            recs, fin = ring.poll(sid)                     # never raises because of what the bytes are
            if recs.size:
                got[i].append(recs.copy())
            done[i] = fin
        ring.pump()
    for sid in sids:
        ring.close_stream(sid)
    return [np.concatenate(g) if g else np.zeros(0, dtype=got_dtype()) for g in got]


def got_dtype():
    from pbs_plus_amd import _lib

    return _lib.RECORD_DTYPE


RING_CASES = [
    # avg, ring options, unit of the crafted stretches, bytes of the mixed stream
    (4096, dict(arena_bytes=40 * (65536 + 256), page_bytes=65536, max_streams=8, sha_cus=4, round_pages=6), 20_000, 900_000),
    (65536, dict(arena_bytes=96 * (262144 + 256), page_bytes=262144, max_streams=8, sha_cus=16, round_pages=16), 150_000,
     12 << 20),
    (4 << 20, dict(arena_bytes=3 << 30, max_streams=8, sha_cus=64, round_pages=64), 3 << 20, 200 << 20),
]


@pytest.mark.parametrize("avg,opt,unit,total", RING_CASES, ids=["avg4096", "avg64Ki", "avg4Mi"])
def test_ring_crafted_dense_streams_are_bit_exact(gpu_lib, O, allp, avg, opt, unit, total):
    """Six streams share the ring and its rounds: a mix of dense and ordinary stretches at unaligned offsets; EVERY position
    a candidate from the first byte to the last; one candidate per 64 bytes behind an ordinary first page; a period that
    cancels (no candidate: max-size cuts); and two ordinary streams. All bit-exact, all pages back, nothing fails."""
    from pbs_plus_amd import PageRing

    cfg = O.new_config(avg)
    onep = D.one_phase_pattern(O, cfg)
    eng = _engine(avg)
    ring = PageRing(eng, **opt)
    page = ring.page_bytes
    datas = [
        D.crafted_stream(O, cfg, total, 7, allp, onep, unit),
        np.tile(allp, (total // 3) // 64 + 1)[: total // 3 + 17],
        np.concatenate([O.fill(page, 9, 0), np.tile(onep if onep is not None else allp, (3 * page + 999) // 64 + 1)[: 3 * page + 999]]),
        np.tile(allp[:32], (2 * page + 5) // 32 + 1)[: 2 * page + 5],
        O.fill(total // 2 + 1234, 70, 0),
        O.fill(total // 4 + 77, 71, 3),
    ]
    got = _feed_host_streams(eng, ring, datas)
    ring.quiesce()
    st = ring.stats()
    for i, (g, d) in enumerate(zip(got, datas)):
        _same(g, O.chunk_and_digest(cfg, d, [(0, d.size)]), (avg, i))
    # the all-candidate stream is cut at the minimum every time (the serial chunker's answer to such bytes)
    assert (got[1]["size"][:-1] == cfg.min).all() and got[1].size == -(-datas[1].size // cfg.min)
    assert st["pages_free"] == st["pages_total"] and st["pages_recycled"] == st["pages_enqueued"], st
    ring.close()
    eng.close()


def test_ring_dense_tiles_whose_cut_lands_in_their_sparse_part(gpu_lib, O, allp):
    """The worst case of the on-demand re-scan: tiles that overflow their slots because ONE short stretch of them is dense,
    while the cut rule's search lands in the ordinary rest — the re-scan has to run to the end of the tile (and into the
    next ones) without a hit, and must neither invent a candidate nor miss the next real one. avg 4096: tiles of 32 KiB
    with 256 slots; each tile gets 600 all-candidate bytes at its front, random bytes behind."""
    from pbs_plus_amd import PageRing

    avg = 4096
    cfg = O.new_config(avg)
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=40 * (65536 + 256), page_bytes=65536, max_streams=4, sha_cus=4, round_pages=6)
    n = 20 * 32768 + 4321
    d = O.fill(n, 321, 0)
    for t in range(0, n - 700, 32768):
        ph = (t // 32768 * 7) % 64
        d[t + 5:t + 605] = np.tile(np.roll(allp, -ph), 11)[:600]
    e = O.fill(n, 322, 0)
    for t in range(0, n - 40000, 32768):                       # ... and at the tiles' ends
        e[t + 32000:t + 32768] = np.tile(allp, 12)[:768]
    got = _feed_host_streams(eng, ring, [d, e])
    ring.quiesce()
    for g, x in zip(got, (d, e)):
        _same(g, O.chunk_and_digest(cfg, x, [(0, x.size)]), "sparse part")
    assert O.candidates(cfg, d[:32768]).size > 256             # the tiles really overflow
    ring.close()
    eng.close()


@pytest.mark.parametrize("avg,total,unit", [(4096, 3_000_000, 20_000), (4 << 20, 160 << 20, 3 << 20)], ids=["avg4096", "avg4Mi"])
def test_batch_path_crafted_dense_input_is_bit_exact(gpu_lib, O, allp, avg, total, unit):
    """pbsgpu_submit_*: the capacity retry stops at one slot per 128 bytes; a batch that still overflows is resolved by the
    same on-demand re-scan (single stream and many segments, host and device submits, a misaligned device pointer) — and
    the engine goes back to the ordinary path for the ordinary batch that follows."""
    cfg = O.new_config(avg)
    onep = D.one_phase_pattern(O, cfg)
    eng = _engine(avg, inflight=2)
    mixed = D.crafted_stream(O, cfg, total, 11, allp, onep, unit)
    pure = np.tile(allp, total // 64 + 1)[:total]
    for name, data in (("mixed", mixed), ("every position", pure)):
        want = O.chunk_and_digest(cfg, data)
        t = eng.submit(data)
        got = eng.collect(t)
        assert records_equal(got, want), (name, describe_mismatch(got, want))
    # many segments (ragged, some inside dense stretches), device-resident, pointer off by 3 bytes
    rng = np.random.default_rng(4)
    cuts = np.sort(rng.choice(np.arange(1, total - 3), size=23, replace=False))
    segs = [(int(a), int(b - a)) for a, b in zip(np.r_[0, cuts], np.r_[cuts, total - 3])]
    buf = eng.alloc(total + 64)
    buf.upload(mixed, offset=3)
    t = eng.submit(buf.ptr + 3, segs, nbytes=total - 3)      # a view 3 bytes into the allocation
    got = eng.collect(t)
    want = O.chunk_and_digest(cfg, mixed[: total - 3], segs)
    assert records_equal(got, want), describe_mismatch(got, want)
    buf.free()
    # an ordinary batch afterwards: bit-exact, and no re-run
    plain = O.fill(total, 5, 0)
    for _ in range(2):
        t = eng.submit(plain)
        got = eng.collect(t)
    assert records_equal(got, O.chunk_and_digest(cfg, plain))
    eng.close()


def test_batch_path_dense_input_with_suggested_boundaries(gpu_lib, O, allp):
    """suggested boundaries (payload chunker) on top of every-position-a-candidate bytes: the walk's two lists and the
    on-demand re-scan together"""
    avg = 4096
    cfg = O.new_config(avg)
    eng = _engine(avg)
    n = 700_000
    data = np.tile(allp, n // 64 + 1)[:n]
    data[300_000:340_000] = O.fill(40_000, 3, 0)
    rng = np.random.default_rng(9)
    sugg = np.sort(rng.choice(np.arange(1, n), size=150, replace=False)).astype(np.uint64)
    got = eng.chunk_and_digest(data, [(0, n)], suggested=[sugg])
    want = O.chunk_and_digest_suggested(cfg, data, [(0, n)], [sugg])
    assert records_equal(got, want), describe_mismatch(got, want)
    eng.close()


def test_payload_stream_takes_crafted_dense_bytes_like_any_others(gpu_lib, O, allp):
    """pbsgpu_stream_* (the WriteEntryReader seam): one archive carries crafted periods in random write sizes, with forced
    cuts (InjectChunks) inside dense stretches; an ordinary archive is written through the same engine at the same time.
    Both bit-exact; no call fails."""
    from pbs_plus_amd import PayloadStream

    avg = 4096
    cfg = O.new_config(avg)
    onep = D.one_phase_pattern(O, cfg)
    eng = _engine(avg)
    bad = D.crafted_stream(O, cfg, 2_400_000, 21, allp, onep, 30_000)
    good = O.fill(2_500_000, 10, 3)
    ko, ok = PayloadStream(eng), PayloadStream(eng)
    rng = np.random.default_rng(6)
    pos, cuts = 0, []
    while pos < bad.size:
        n = int(rng.integers(1, 90_000))
        ko.write(bad[pos:pos + n])
        ok.write(good[pos:pos + n])
        pos += n
        if rng.random() < 0.15 and pos < bad.size:
            ko.inject(0)                                       # a forced cut at the current position
            cuts.append(pos)
    ok.write(good[pos:])
    ko.finish()
    ok.finish()
    edges = [0] + cuts + [bad.size]
    segs = [(a, b - a) for a, b in zip(edges[:-1], edges[1:]) if b > a]
    want = O.chunk_and_digest(cfg, bad, segs)
    got = ko.poll()
    wend = np.asarray([a for a, _ in segs], dtype=np.uint64)[want["segment"]] + want["end"]
    assert got.size == want.size and np.array_equal(got["end"], wend) and np.array_equal(got["digest"], want["digest"])
    g2 = ok.poll()
    w2 = O.chunk_and_digest(cfg, good, [(0, good.size)])
    assert np.array_equal(g2["end"], w2["end"]) and np.array_equal(g2["digest"], w2["digest"])
    ko.close()
    ok.close()
    eng.close()
