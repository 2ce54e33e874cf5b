"""Parity of the HIP engine (through the C ABI) against the CPU oracle — bit-exact, integer
work. Mirrors how the reference's tests drive the module: buzhash.NewConfig(4096) stores
(internal/pxarmount/commit_walk_test.go:25) and the production avg 4 << 20
(commit_orchestrate.go:144). Edge cases: empty / tiny / ragged segments, min == window,
zero runs (max-size cuts), misaligned buffers, dense candidates."""
import hashlib

import numpy as np
import pytest

from helpers import (describe_mismatch, golden_case_data, golden_files, golden_records, golden_suggested_data, load_golden,
                     records_equal)

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(gpu_lib):
    from pbs_plus_amd import Engine, buzhash

    cache = {}

    def get(avg):
        if avg not in cache:
            cache[avg] = Engine(buzhash.NewConfig(avg), device=0, inflight=2)
        return cache[avg]

    yield get
    for e in cache.values():
        e.close()


def dev_fill(eng, segs):
    """device buffer holding the concatenated synthetic segments [(seed, kind, n)], 8-byte aligned starts"""
    table, off = [], 0
    for _, _, n in segs:
        table.append((off, n))
        off += (n + 7) & ~7
    buf = eng.alloc(max(off, 8))
    for (seed, kind, n), (o, _) in zip(segs, table):
        if n:
            eng.fill(buf.ptr + o, n, seed, kind)
    return buf, table, off


def host_twin(O, segs, table, total):
    out = np.zeros(max(total, 8), dtype=np.uint8)
    for (seed, kind, n), (o, _) in zip(segs, table):
        if n:
            out[o:o + n] = O.fill(n, seed, kind)
    return out


def test_device_fill_matches_oracle_fill(engines, O):
    eng = engines(4096)
    for kind in range(5):
        n = (1 << 20) + 13
        buf = eng.alloc(n + 8)
        eng.fill(buf.ptr, n, 42 + kind, kind, stream_off=4096)
        got = buf.download(0, n)
        assert np.array_equal(got, O.fill(n, 42 + kind, kind, stream_off=4096)), kind
        buf.free()


@pytest.mark.parametrize("avg,n,kind", [(4096, 1 << 20, 0), (4096, (3 << 20) + 17, 3), (256, 300_000, 0),
                                        (65536, 16 << 20, 0), (4 << 20, 64 << 20, 0), (4096, 700_001, 2)])
def test_candidates_match_oracle(engines, O, avg, n, kind):
    """Kernel K1 alone: every window-hash candidate, ascending."""
    eng = engines(avg)
    buf, table, total = dev_fill(eng, [(7 + avg % 97, kind, n)])
    got = eng.candidates(buf, n)
    want = O.candidates(O.new_config(avg), O.fill(n, 7 + avg % 97, kind))
    buf.free()
    assert got.size == want.size, (got.size, want.size, got[:5], want[:5])
    assert np.array_equal(got, want)


@pytest.mark.parametrize("lead", [1, 3, 8, 15])
def test_candidates_misaligned_base(engines, O, lead):
    eng = engines(4096)
    n = 500_000
    host = O.fill(n, 5)
    buf = eng.alloc(n + 64)
    buf.upload(host, offset=lead)
    got = eng.candidates(buf.ptr + lead, n)
    buf.free()
    assert np.array_equal(got, O.candidates(O.new_config(4096), host))


CASES = [
    ("single_4k", 4096, [(1, 0, 1 << 20)]),
    ("ragged_4k", 4096, [(2, 0, 300_000), (3, 1, 70_000), (4, 2, 200_001), (5, 3, 400_003), (6, 0, 17), (7, 0, 0),
                         (8, 0, 1024), (9, 0, 1089), (10, 0, 63), (11, 0, 64), (12, 0, 65), (13, 0, 1)]),
    ("min_eq_window", 256, [(14, 0, 200_000), (15, 1, 5000)]),
    ("avg64k", 65536, [(16, 0, 12 << 20), (17, 3, 5 << 20)]),
    ("avg4m_prod", 4 << 20, [(18, 0, 96 << 20)]),
    ("avg4m_zero_max_cuts", 4 << 20, [(19, 1, (40 << 20) + 5), (20, 3, 40 << 20)]),
    ("many_small", 4096, [(100 + i, i % 4, 1000 + 997 * i) for i in range(150)]),
]


@pytest.mark.parametrize("name,avg,segs", CASES, ids=[c[0] for c in CASES])
def test_chunk_and_digest_matches_oracle(engines, O, name, avg, segs):
    eng = engines(avg)
    buf, table, total = dev_fill(eng, segs)
    got = eng.chunk_and_digest(buf, table, nbytes=max(total, 8))
    want = O.chunk_and_digest(O.new_config(avg), host_twin(O, segs, table, total), table)
    buf.free()
    assert records_equal(got, want), describe_mismatch(got, want)


def test_host_submit_equals_device_submit(engines, O):
    eng = engines(4096)
    data = O.fill(2_000_003, 31)
    got = eng.chunk_and_digest(data, [(0, 1_000_000), (1_000_001, 1_000_002)])
    want = O.chunk_and_digest(O.new_config(4096), data, [(0, 1_000_000), (1_000_001, 1_000_002)])
    assert records_equal(got, want), describe_mismatch(got, want)
    # default: the whole buffer is one stream
    got = eng.chunk_and_digest(data)
    want = O.chunk_and_digest(O.new_config(4096), data)
    assert records_equal(got, want), describe_mismatch(got, want)


def test_misaligned_chunk_starts_hash_correctly(engines):
    """Chunk starts land on arbitrary byte offsets: the per-lane funnel loads must handle all
    four alignments; digests checked against hashlib independently of the oracle."""
    eng = engines(4096)
    rng = np.random.default_rng(9)
    data = rng.integers(0, 256, 400_000, dtype=np.uint8)
    for lead in (0, 1, 2, 3, 5):
        buf = eng.alloc(data.size + 64)
        buf.upload(data, offset=lead)
        recs = eng.chunk_and_digest(buf.ptr + lead, [(0, data.size)], nbytes=data.size)
        buf.free()
        start = 0
        for r in recs:
            assert bytes(r["digest"]) == hashlib.sha256(data[start:int(r["end"])].tobytes()).digest(), (lead, start)
            start = int(r["end"])
        assert start == data.size


@pytest.mark.parametrize("fixture", golden_files())
def test_golden_fixture_on_gpu(engines, O, fixture):
    """every tests/golden/chunks_*.json through the C ABI: the committed fixture, and the REAL Go module's vectors
    (chunks_go.json, `make golden-go`) as soon as a maintainer has produced them — records and, for schema v2, the
    suggested-boundary cuts at every reader-buffer size"""
    g = load_golden(fixture)
    from pbs_plus_amd import RECORD_DTYPE

    for case in g["cases"]:
        eng = engines(case["avg"])
        data, table = golden_case_data(O, case)   # golden segments are packed back to back (no alignment padding): built on the host
        got = eng.chunk_and_digest(data, table if len(table) > 1 else None)
        want = golden_records(case, RECORD_DTYPE)
        assert records_equal(got, want), fixture + " " + case["name"] + "\n" + describe_mismatch(got, want)
    for case in g.get("suggested", []):
        eng = engines(case["avg"])
        data = golden_suggested_data(O, case)
        eng.set_suggested_feed(int(case["feed"]), False)
        try:
            got = eng.chunk_and_digest(data, [(0, data.size)], suggested=[sorted(case["suggested"])])
        finally:
            eng.set_suggested_feed(1, False)
        assert got["end"].tolist() == [int(e) for e in case["ends"]], (fixture, case["name"], case["feed"])


def test_sha256_many_all_padding_lengths(engines):
    """verification.HashFile for many files at once (handler.go:36-68): every padding case."""
    eng = engines(4096)
    rng = np.random.default_rng(4)
    lens = list(range(0, 200)) + [255, 256, 257, 4095, 4096, 65537, 1_000_003]
    blob = rng.integers(0, 256, sum(lens) + 16, dtype=np.uint8)
    segs, off = [], 0
    for n in lens:
        segs.append((off, n))
        off += n
    got = eng.sha256_many(blob, segs)
    for (o, n), d in zip(segs, got):
        assert bytes(d) == hashlib.sha256(blob[o:o + n].tobytes()).digest(), n


def test_pipelined_tickets(engines, O):
    """Two batches in flight on separate slots/streams give the same answers as one at a time."""
    eng = engines(65536)
    a, b = O.fill(9 << 20, 61), O.fill(7 << 20, 62, 3)
    ta, tb = eng.submit(a), eng.submit(b)
    from pbs_plus_amd import PbsGpuError

    with pytest.raises(PbsGpuError):
        eng.submit(a)  # both slots busy
    rb, ra = eng.collect(tb), eng.collect(ta)
    cfg = O.new_config(65536)
    assert records_equal(ra, O.chunk_and_digest(cfg, a)) and records_equal(rb, O.chunk_and_digest(cfg, b))
    with pytest.raises(PbsGpuError):
        eng.collect(ta)


def test_dense_candidates_force_capacity_retry(engines, O):
    """A 64-byte-periodic input makes the window hash periodic: either no candidates or a hit
    every period. Search a period that hits, then check the density-retry path."""
    eng = engines(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(123)
    for _ in range(4000):
        block = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(block, 4)).size:
            break
    else:
        pytest.skip("no dense pattern found")
    data = np.tile(block, 20_000)  # 1.28 MB, a candidate at least every 64 bytes
    assert O.candidates(cfg, data).size >= 19_000
    t = eng.submit(data)
    got = eng.collect(t)
    want = O.chunk_and_digest(cfg, data)
    assert records_equal(got, want), describe_mismatch(got, want)


def test_payload_stream_equals_batch(engines, O):
    """PayloadStream (WriteEntryReader seam): windows + carry-over reproduce one-shot cutting."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    cfg = O.new_config(4096)
    data = O.fill(1_300_007, 71, 3)
    ps = PayloadStream(eng, window_bytes=1 << 16)  # small windows: many flushes with carry
    rng = np.random.default_rng(2)
    pos = 0
    while pos < data.size:
        n = int(rng.integers(1, 90_000))
        ps.write(data[pos:pos + n])
        pos += n
    ps.finish()
    got = ps.poll()
    want = O.chunk_and_digest(cfg, data)
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    assert ps.position() == data.size
    ps.close()


def test_payload_stream_inject_forces_cut(engines, O):
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    cfg = O.new_config(4096)
    a, b = O.fill(250_000, 81), O.fill(99_999, 82)
    ps = PayloadStream(eng, window_bytes=1 << 17)
    ps.write(a)
    ps.inject(5_000_000)  # known chunks spliced in: open chunk is flushed, offsets skip ahead
    ps.write(b)
    ps.finish()
    got = ps.poll()
    ra, rb = O.chunk_and_digest(cfg, a), O.chunk_and_digest(cfg, b)
    assert np.array_equal(got["end"], np.concatenate([ra["end"], rb["end"] + np.uint64(a.size + 5_000_000)]))
    assert np.array_equal(got["digest"], np.concatenate([ra["digest"], rb["digest"]]))
    assert got["segment"].tolist() == [0] * ra.size + [1] * rb.size
    ps.close()


def test_chunker_scan_semantics(engines, O):
    """Upstream `scan`: whole-buffer feeds and small feeds give the oracle's boundaries."""
    from pbs_plus_amd import Chunker

    eng = engines(4096)
    cfg = O.new_config(4096)
    data = O.fill(600_000, 91)
    want = O.chunk_stream(cfg, data)
    for step in (data.size, 50_000, 4097):
        ch = Chunker(eng)
        ends, pos = [], 0
        while pos < data.size:
            n = min(step, data.size - pos)
            off = 0
            while off < n:
                k = ch.scan(data[pos + off:pos + n])
                if k == 0:
                    break
                off += k
                ends.append(pos + off)
            pos += n
        if not ends or ends[-1] != data.size:
            ends.append(data.size)
        assert np.array_equal(np.asarray(ends, dtype=np.uint64), want), step
        ch.close()


def test_dedup_and_didx(engines, O):
    from pbs_plus_amd.engine import didx_decode

    eng = engines(4096)
    base = O.fill(400_000, 95)
    data = np.concatenate([base, O.fill(100_000, 96), base])  # third segment duplicates the first
    segs = [(0, 400_000), (400_000, 100_000), (500_000, 400_000)]
    recs = eng.chunk_and_digest(data, segs)
    dup, stats = eng.dedup(recs)
    n0 = int((recs["segment"] == 0).sum())
    seen, want_dup = set(), []
    for r in recs:
        d = bytes(r["digest"])
        want_dup.append(1 if d in seen else 0)
        seen.add(d)
    assert dup.tolist() == want_dup
    assert sum(want_dup) >= n0 - 1
    assert stats["nrecords"] == recs.size and stats["nunique"] == len(seen)
    assert stats["total_bytes"] == data.size
    assert stats["unique_bytes"] == int(recs["size"][dup == 0].sum())
    # dynamic index of the first segment
    r0 = recs[recs["segment"] == 0]
    blob = eng.didx_encode(r0, uuid=bytes(range(16)), ctime=1_700_000_123)
    back, ctime, csum = didx_decode(blob)
    assert ctime == 1_700_000_123 and np.array_equal(back["end"], r0["end"]) and np.array_equal(back["digest"], r0["digest"])
    assert csum == hashlib.sha256(blob[4096:]).digest()
    assert blob[8:24] == bytes(range(16))


def test_many_files_mixed_entropy_batch(engines, O):
    """BASELINE.json configs[2] shape (many files packed into one batch, entropy class by file
    index), scaled to what the oracle finishes in seconds: 96 files x 6 MiB, production avg."""
    eng = engines(4 << 20)
    segs = [(300 + i, i % 4, (6 << 20) + 4099 * i) for i in range(96)]
    buf, table, total = dev_fill(eng, segs)
    got = eng.chunk_and_digest(buf, table, nbytes=total)
    want = O.chunk_and_digest(O.new_config(4 << 20), host_twin(O, segs, table, total), table)
    buf.free()
    assert records_equal(got, want), describe_mismatch(got, want)
    assert got.size >= 96 and set(got["segment"].tolist()) == set(range(96))


def test_full_size_config2_properties(engines, O):
    """BASELINE.json configs[1] at FULL size (single 64 GiB stream, avg 4 MiB): properties that do
    not need a 64 GiB oracle run — (1) ends strictly increasing and covering the stream, sizes in
    [min, max] except the tail; (2) count in the statistical band; (3) causality: the records before
    offset P equal the oracle's on the first P bytes; (4) sampled chunk digests equal hashlib on the
    downloaded bytes; (5) max-size cuts only where no candidate intervenes is implied by (3)."""
    eng = engines(4 << 20)
    n = 64 << 30
    try:
        buf = eng.alloc(n)
    except Exception as exc:  # pragma: no cover - only on boxes with less free HBM
        pytest.skip(f"cannot allocate 64 GiB: {exc}")
    eng.fill(buf.ptr, n, seed=2, kind=0)
    recs = eng.chunk_and_digest(buf, None, nbytes=n)
    ends = recs["end"].astype(np.int64)
    sizes = recs["size"].astype(np.int64)
    assert ends[-1] == n and np.all(np.diff(ends) > 0)
    assert np.array_equal(np.diff(np.concatenate([[0], ends])), sizes)
    assert sizes[:-1].min() >= 1 << 20 and sizes.max() <= 16 << 20
    assert 16_500 < recs.size < 19_000  # mean chunk ~3.7 MiB
    assert (sizes == 16 << 20).sum() >= 20  # forced cuts exist (P ~ 0.4 %)
    # causality vs the oracle on a 256 MiB prefix
    P = 256 << 20
    want = O.chunk_and_digest(O.new_config(4 << 20), O.fill(P, 2, 0))
    k = want.size - 1
    assert k > 50
    assert np.array_equal(recs["end"][:k], want["end"][:k]) and np.array_equal(recs["digest"][:k], want["digest"][:k])
    # sampled digests across the whole stream, independent of the oracle
    rng = np.random.default_rng(0)
    for i in rng.choice(recs.size, 12, replace=False):
        start = int(ends[i] - sizes[i])
        data = buf.download(start, int(sizes[i]))
        assert bytes(recs["digest"][i]) == hashlib.sha256(data.tobytes()).digest(), i
    # oracle parity ACROSS the stream (restart points: a cut resets the chunker, so the records behind any cut depend
    # only on the bytes behind it): 48 points spread over the 64 GiB incl. the first cuts beyond 2^32 and 2^35 and the
    # span that reaches the final chunk in the last scan tile — cuts AND digests vs the oracle on downloaded bytes
    from oracle import restart_check as RC
    r = RC.check_batch(lambda off, m: buf.download(off, m), None, recs, 4 << 20, nbytes=n, k=48, span=64 << 20, threads=16)
    assert r["ok"], r
    assert r["records_checked"] > 400 and r["max_offset"] == n and r["points"] >= 48
    # raw candidates of a window that lies beyond 60 GiB, vs the oracle's candidate list of the downloaded bytes
    woff, wlen = (61 << 30) + 4096 + 17, (96 << 20) + 5
    got_c = eng.candidates(buf.ptr + woff, wlen)
    want_c = O.candidates(O.new_config(4 << 20), buf.download(woff, wlen))
    assert np.array_equal(got_c, want_c) and want_c.size >= 5, (got_c.size, want_c.size)
    buf.free()


def test_payload_pack_layout_and_chunking(engines, O):
    """(f)-3 payload-stream assembly: markers + 16-byte {type, 16+len} headers + bodies, then the
    packed stream is cut as ONE stream (chunks span files) and must equal the oracle on the same bytes.
    Header arithmetic mirrors commit_types.go:24-32 (rangeEnd = offset + FileSize + HeaderSize)."""
    eng = engines(4096)
    rng = np.random.default_rng(17)
    lens = [0, 1, 15, 16, 17, 4096, 100_003, 1 << 20, 333, 2_000_001]
    offs, pos = [], 3  # odd source offsets: misaligned gather
    for n in lens:
        offs.append(pos)
        pos += n + int(rng.integers(0, 7))
    src_host = rng.integers(0, 256, pos + 64, dtype=np.uint8)
    src = eng.alloc(src_host.size)
    src.upload(src_host)
    files = list(zip(offs, lens))
    total = 32 + sum(16 + n for n in lens)
    dst = eng.alloc(total + 5)
    out_len, hdr_offs = eng.payload_pack(src, files, dst)
    assert out_len == total
    got = dst.download(0, total)
    PAYLOAD, START, TAIL = 0x28147a1b0b7c1a25, 0x834c68c2194a4ed2, 0x6c72b78b984c81b5
    want = bytearray(START.to_bytes(8, "little") + (16).to_bytes(8, "little"))
    exp_offs = []
    for o, n in files:
        exp_offs.append(len(want))
        want += PAYLOAD.to_bytes(8, "little") + (16 + n).to_bytes(8, "little") + src_host[o:o + n].tobytes()
    want += TAIL.to_bytes(8, "little") + (16).to_bytes(8, "little")
    assert got.tobytes() == bytes(want)
    assert hdr_offs.tolist() == exp_offs
    # reference arithmetic: next file's header starts at rangeEnd = offset + FileSize + HeaderSize
    for (o, n), a, b in zip(files[:-1], exp_offs[:-1], exp_offs[1:]):
        assert b == a + n + 16
    recs = eng.chunk_and_digest(dst, [(0, total)], nbytes=total)
    wrecs = O.chunk_and_digest(O.new_config(4096), np.frombuffer(bytes(want), dtype=np.uint8))
    assert records_equal(recs, wrecs), describe_mismatch(recs, wrecs)
    src.free()
    dst.free()


def test_xxh3_many_matches_xxhash_library(engines):
    """XXH3-64 of many files at once (the hash writeBackedFile tees through, commit_reuse.go:450-461),
    checked against the independent xxhash C library on every length class and odd alignments."""
    import xxhash

    eng = engines(4096)
    rng = np.random.default_rng(21)
    lens = list(range(0, 260)) + [511, 512, 1023, 1024, 1025, 2047, 2048, 2049, 4097, 65_537, 1_000_003, 3 << 20]
    blob = rng.integers(0, 256, sum(lens) + len(lens) * 3 + 64, dtype=np.uint8)
    segs, off = [], 1
    for n in lens:
        segs.append((off, n))
        off += n + (n % 3)  # ragged, unaligned starts
    got = eng.xxh3_many(blob, segs)
    for (o, n), g in zip(segs, got):
        assert int(g) == xxhash.xxh3_64_intdigest(blob[o:o + n].tobytes()), n
    # device-resident variant
    buf = eng.alloc(blob.size)
    buf.upload(blob)
    got2 = eng.xxh3_many(buf, segs)
    buf.free()
    assert np.array_equal(got, got2)


def test_alternate_kernels_stay_bit_exact(gpu_lib):
    """The alternative hash kernels of the batch path (single-wave lanes, express pairs) are an ENGINE option since round 6
    (pbsgpu_engine_options::sha_form; the PBSGPU_SHA_MODE variable of rounds 1-5 still works as a debug override): both ways
    of selecting them, a small parity check each. (The LDS-tiled and per-lane streaming scan kernels are gone.)"""
    import subprocess
    import sys

    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from pbs_plus_amd import Engine, buzhash\n"
        "from tests.helpers import records_equal\n"
        "import os\n"
        "opt = dict(sha_form=int(os.environ['SHA_FORM'])) if 'SHA_FORM' in os.environ else {}\n"
        "for avg, n, kind in ((4096, 3_000_001, 3), (4 << 20, 80 << 20, 0)):\n"
        "    eng = Engine(buzhash.NewConfig(avg), **opt)\n"
        "    data = O.fill(n, 5, kind)\n"
        "    assert records_equal(eng.chunk_and_digest(data), O.chunk_and_digest(O.new_config(avg), data)), avg\n"
        "    eng.close()\n"
        "print('alt-ok')\n" % __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__))))
    for alt in (dict(PBSGPU_SHA_MODE="lane"), dict(SHA_FORM="1"), dict(SHA_FORM="2")):
        env = dict(__import__("os").environ, **alt)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=120)
        assert "alt-ok" in out.stdout, str(alt) + out.stdout + out.stderr


def test_engine_is_safe_from_many_threads(engines, O):
    """cgo calls arrive on arbitrary OS threads: 4 threads hammer one engine (each its own data);
    E_BUSY is a legal answer when both slots are taken, wrong results are not."""
    import threading

    from pbs_plus_amd import PbsGpuError, _lib

    eng = engines(4096)
    cfg = O.new_config(4096)
    datas = [O.fill(700_000 + 1111 * i, 400 + i, i % 4) for i in range(4)]
    wants = [O.chunk_and_digest(cfg, d) for d in datas]
    errors = []

    def work(i):
        try:
            for _ in range(6):
                while True:
                    try:
                        got = eng.chunk_and_digest(datas[i])
                        break
                    except PbsGpuError as exc:
                        if exc.status != _lib.E_BUSY:
                            raise
                if not records_equal(got, wants[i]):
                    errors.append((i, "mismatch"))
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    ts = [threading.Thread(target=work, args=(i,)) for i in range(4)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors


def test_payload_stream_zero_copy_feed(engines, O):
    """reserve/commit (the io.ReaderFrom form of the WriteEntryReader seam) == write()."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    data = O.fill(900_001, 77, 0)
    ps = PayloadStream(eng, window_bytes=1 << 17)
    off = 0
    rng = np.random.default_rng(3)
    while off < data.size:
        buf = ps.reserve()
        k = int(min(buf.size, data.size - off, rng.integers(1, 70_000)))
        buf[:k] = data[off:off + k]
        ps.commit(k)
        off += k
    ps.finish()
    got = ps.poll()
    want = O.chunk_and_digest(O.new_config(4096), data)
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    ps.close()


def test_incremental_rechunk_after_edits_keeps_boundaries_local(engines, O):
    """BASELINE.json configs[4] shape (re-chunk after ~2 % random edits, boundary-shift stress), scaled:
    64 MiB corpus, avg 64 KiB; extents of log-uniform length, 1/3 overwrite, 1/3 insert, 1/3 delete.
    (1) the edited stream is bit-exact vs the oracle; (2) content-defined cuts resynchronise, so the
    share of chunks whose digest already existed stays high (device digest-set dedup)."""
    eng = engines(65536)
    cfg = O.new_config(65536)
    rng = np.random.default_rng(5)
    old = O.fill(64 << 20, 500, 0)
    pieces, pos, edited = [], 0, 0
    target = int(0.02 * old.size)
    cuts = np.sort(rng.choice(old.size - (1 << 20), 40, replace=False))
    for c in cuts:
        if c < pos:
            continue
        if edited >= target:
            break
        ln = int(np.exp(rng.uniform(np.log(4096), np.log(256 << 10))))
        kind = int(rng.integers(0, 3))
        pieces.append(old[pos:c])
        if kind == 0:    # overwrite
            pieces.append(rng.integers(0, 256, ln, dtype=np.uint8))
            pos = c + ln
        elif kind == 1:  # insert
            pieces.append(rng.integers(0, 256, ln, dtype=np.uint8))
            pos = c
        else:            # delete
            pos = c + ln
        edited += ln
    pieces.append(old[pos:])
    new = np.concatenate(pieces)
    assert 0.005 * old.size < edited < 0.06 * old.size
    r_old = eng.chunk_and_digest(old)
    r_new = eng.chunk_and_digest(new)
    want = O.chunk_and_digest(cfg, new)
    assert records_equal(r_new, want), describe_mismatch(r_new, want)
    both = np.concatenate([r_old, r_new])
    dup, stats = eng.dedup(both)
    reused = int(dup[r_old.size:].sum())
    assert reused / r_new.size > 0.85, (reused, r_new.size)
    assert stats["nunique"] < r_old.size + 0.15 * r_new.size


def test_dedup_digest_prefix_collisions_are_not_duplicates(engines):
    """The device dedup sorts by the first 8 digest bytes: records that share that prefix but differ
    later are NOT duplicates (domain 'collision' case), true duplicates inside the same run are."""
    from pbs_plus_amd import RECORD_DTYPE

    eng = engines(4096)
    rng = np.random.default_rng(6)
    recs = np.zeros(40, dtype=RECORD_DTYPE)
    recs["size"] = rng.integers(1, 1 << 20, 40)
    recs["digest"] = rng.integers(0, 256, (40, 32), dtype=np.uint8)
    recs["digest"][:24, :8] = recs["digest"][0, :8]          # 24 records share one 8-byte prefix ...
    recs["digest"][10] = recs["digest"][3]                    # ... of which two pairs are real duplicates
    recs["digest"][20] = recs["digest"][7]
    recs["digest"][35] = recs["digest"][30]                   # and one duplicate pair with a unique prefix
    dup, stats = eng.dedup(recs)
    want = np.zeros(40, dtype=np.uint8)
    want[[10, 20, 35]] = 1
    assert dup.tolist() == want.tolist()
    assert stats["nunique"] == 37 and stats["unique_bytes"] == int(recs["size"][want == 0].sum())


def test_cpp_mirror_writer_matches_oracle(gpu_lib, O, tmp_path):
    """include/pbsgpu.hpp on the GPU: PayloadWriter.WriteEntry / WriteEntryReader / Finish and the
    DynamicIndexWriter round trip, driven the way commit_walk_test.go drives the Go module."""
    import subprocess

    root = __import__("os").path.dirname(__import__("os").path.dirname(__import__("os").path.abspath(__file__)))
    exe = str(tmp_path / "test_cpp_writer")
    libdir = __import__("os").path.join(root, "pbs_plus_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", __import__("os").path.join(root, "tests", "native", "test_cpp_writer.cpp"),
                    "-L" + libdir, "-lpbsgpu", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    n = 700_001
    out = subprocess.run([exe, str(n)], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpp-writer-ok" in out.stdout, out.stdout + out.stderr
    lines = [l.split() for l in out.stdout.splitlines() if l and not l.startswith("cpp-")]
    want = O.chunk_and_digest(O.new_config(4096), O.fill(n, 77, 0))
    assert len(lines) == want.size
    for (end, size, dig), w in zip(lines, want):
        assert int(end) == int(w["end"]) and int(size) == int(w["size"]) and dig == bytes(w["digest"]).hex()


def test_tiny_entries_never_reach_a_content_cut(engines):
    """The reference's own store round-trips write files of <= 17 bytes (commit_walk_test.go:21-147):
    far below min chunk, so no content-defined cut is ever taken — the archive payload is ONE chunk whose
    digest is the SHA-256 of the concatenation. Same through PayloadStream, entry by entry."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    ps = PayloadStream(eng, window_bytes=1 << 16)
    entries = [b"hello world", b"", b"a", b"0123456789abcdef!", b"nested/file", b"\x00" * 17, b"tail"]
    for e in entries:
        ps.write(e)
    ps.finish()
    recs = ps.poll()
    blob = b"".join(entries)
    assert recs.size == 1 and int(recs["end"][0]) == len(blob) == ps.position()
    assert bytes(recs["digest"][0]) == hashlib.sha256(blob).digest()
    ps.close()
    # and an empty archive produces no chunk at all
    ps = PayloadStream(eng)
    ps.finish()
    assert ps.poll().size == 0
    ps.close()


def test_device_entry_points_reject_host_pointers(engines):
    """A host buffer handed to a *_device entry point (an easy cgo mistake) is refused, not dereferenced."""
    import ctypes as C

    from pbs_plus_amd import _lib

    eng = engines(4096)
    host = np.zeros(1 << 16, dtype=np.uint8)
    t = C.c_uint64()
    st = eng._L.pbsgpu_submit_device(eng._h, host.ctypes.data, host.size, None, 0, C.byref(t))
    assert st == _lib.E_INVALID
    n = C.c_uint64()
    assert eng._L.pbsgpu_candidates_device(eng._h, host.ctypes.data, host.size, None, 0, C.byref(n)) == _lib.E_INVALID
    # the engine is still usable afterwards
    assert eng.chunk_and_digest(host).size == 4  # 64 KiB of zeros, avg 4 KiB -> four max-size (16 KiB) chunks


def test_cross_file_duplication_fraction(engines, O):
    """BASELINE.json configs[3] shape, scaled: a corpus of files of which 40 % are exact copies of an
    earlier file (seeded permutation). Whole-file copies yield identical chunk sequences, so the device
    digest-set reduce must report exactly the copied files' bytes as duplicates."""
    eng = engines(65536)
    rng = np.random.default_rng(4)
    nfiles, flen = 50, 1 << 20
    originals = {}
    src_of = []
    for i in range(nfiles):
        if i >= 5 and rng.random() < 0.4:
            src_of.append(int(rng.integers(0, i)))
        else:
            src_of.append(i)
    # resolve copy-of-copy chains to the original
    for i in range(nfiles):
        while src_of[src_of[i]] != src_of[i]:
            src_of[i] = src_of[src_of[i]]
    parts = []
    for i in range(nfiles):
        s = src_of[i]
        if s not in originals:
            originals[s] = O.fill(flen + 4099 * (s % 7), 600 + s, 0)
        parts.append(originals[s])
    offs, pos = [], 0
    for p in parts:
        offs.append(pos)
        pos += p.size
    data = np.concatenate(parts)
    segs = [(o, p.size) for o, p in zip(offs, parts)]
    recs = eng.chunk_and_digest(data, segs)
    dup, stats = eng.dedup(recs)
    dup_bytes_expected = sum(p.size for i, p in enumerate(parts) if src_of[i] != i)
    assert stats["total_bytes"] == data.size
    assert stats["total_bytes"] - stats["unique_bytes"] == dup_bytes_expected
    assert 0.2 < dup_bytes_expected / data.size < 0.6
    # duplicates are flagged on the LATER occurrence only
    first_seen = {}
    for r, d in zip(recs, dup):
        key = bytes(r["digest"])
        assert int(d) == (1 if key in first_seen else 0)
        first_seen.setdefault(key, int(r["segment"]))


def _split_worker(rank, ws, port, q):
    import os
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    import torch.distributed as dist

    from oracle import oracle as O
    from pbs_plus_amd import Engine, buzhash
    from pbs_plus_amd.dist import split_plan, split_stream_chunk_and_digest

    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=ws)
    try:
        T = (40 << 20) + 12345
        cfg = buzhash.NewConfig(65536)
        eng = Engine(cfg, device=0)  # both ranks share the one GPU of the test box
        a, b, lo, hi = split_plan(T, ws, cfg.MaxSize)[rank]
        whole = O.fill(T, 4242, 3)
        buf = eng.alloc(hi - lo + 8)
        buf.upload(whole[lo:hi])
        recs = split_stream_chunk_and_digest(eng, buf, T)
        q.put((rank, recs.tobytes()))
        buf.free()
        eng.close()
    finally:
        dist.destroy_process_group()


def test_single_stream_split_over_two_ranks(gpu_lib, O):
    """one stream split over ranks (63-byte halo, max-size right overlap, gathered candidate
    lists, identical resolve everywhere, per-rank hashing) equals the oracle on the whole stream. Two
    processes over gloo share the single GPU of the test box; on an 8-GPU node the same code runs over RCCL."""
    import socket

    import torch.multiprocessing as mp

    from pbs_plus_amd import RECORD_DTYPE

    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    ctx = mp.get_context("spawn")
    q = ctx.Queue()
    procs = [ctx.Process(target=_split_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    got = sorted(q.get(timeout=150) for _ in range(2))
    for p in procs:
        p.join(timeout=60)
        assert p.exitcode == 0
    r0 = np.frombuffer(got[0][1], dtype=RECORD_DTYPE)
    r1 = np.frombuffer(got[1][1], dtype=RECORD_DTYPE)
    assert r0.tobytes() == r1.tobytes()
    T = (40 << 20) + 12345
    want = O.chunk_and_digest(O.new_config(65536), O.fill(T, 4242, 3))
    assert records_equal(r0, want), describe_mismatch(r0, want)


@pytest.mark.gpu
def test_ticket_done_lets_the_host_collect_out_of_order(engines, O):
    """pbsgpu_ticket_done: non-blocking completion query, so a host with several tickets in flight can collect
    whichever finishes first (bench.py --collect any). A small ticket submitted after a large one is collected
    first; both equal the oracle."""
    import time

    eng = engines(65536)
    big, small = O.fill(96 << 20, 91, 0), O.fill(1 << 20, 92, 0)
    t_big = eng.submit(big)
    t_small = eng.submit(small)
    deadline = time.time() + 60
    while not eng.done(t_small):
        assert time.time() < deadline, "ticket never reported done"
        time.sleep(0.001)
    got_small = eng.collect(t_small)
    got_big = eng.collect(t_big)
    cfg = O.new_config(65536)
    assert records_equal(got_small, O.chunk_and_digest(cfg, small))
    assert records_equal(got_big, O.chunk_and_digest(cfg, big))
    with pytest.raises(Exception):
        eng.done(t_small)  # released ticket
