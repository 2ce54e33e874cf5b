"""Round-2 GPU parity tests (through the C ABI, against the CPU oracle, bit-exact): suggested boundaries
(payload chunker), PayloadPosition with injected bytes, many streams / threads sharing one engine without
E_BUSY, >4096 segments per submit, 16 tickets in flight, dense-tile compaction, the piece-table gather, the
shared hash jobs of the streaming writer at production chunk size, and handle lifetime."""
import threading

import numpy as np
import pytest

from helpers import describe_mismatch, records_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def engines(gpu_lib):
    from pbs_plus_amd import Engine, buzhash

    cache = {}

    def get(avg, inflight=2):
        if (avg, inflight) not in cache:
            cache[(avg, inflight)] = Engine(buzhash.NewConfig(avg), device=0, inflight=inflight)
        return cache[(avg, inflight)]

    yield get
    for e in cache.values():
        e.close()


def random_suggestions(rng, seg_len, cfg, n):
    """ascending suggested boundaries that hit every branch: closer than min, exactly min / max apart, duplicates,
    beyond max, at and past the segment end"""
    pts = set(int(x) for x in rng.integers(1, max(2, seg_len), n))
    base = int(rng.integers(0, max(1, seg_len // 2)))
    pts |= {base + cfg.min, base + cfg.min - 1, base + cfg.max, base + cfg.max + 1, seg_len, seg_len + 5, 64, 65}
    out = sorted(p for p in pts if p > 0)
    return out + out[:3]  # a few duplicates; re-sorted below


@pytest.mark.parametrize("avg", [256, 4096, 65536])
def test_suggested_boundaries_match_the_payload_chunker(engines, O, avg):
    """pbsgpu_submit_*_suggested == oracle_payload_chunker_scan fed byte by byte (SURVEY.md Appendix A note 3)."""
    eng = engines(avg)
    cfg = O.new_config(avg)
    rng = np.random.default_rng(avg)
    lens = [avg * 37 + 13, 0, avg * 5, 63, avg * 61 + 1, avg // 2]
    kinds = [0, 0, 1, 0, 3, 0]          # a zero run: only max-size cuts unless a suggestion intervenes
    segs, off = [], 0
    for n in lens:
        segs.append((off, n))
        off += (n + 7) & ~7
    data = np.zeros(max(off, 8), dtype=np.uint8)
    for (o, n), k, i in zip(segs, kinds, range(len(segs))):
        if n:
            data[o:o + n] = O.fill(n, 900 + i, k)
    sugg = [sorted(random_suggestions(rng, n, cfg, 40)) for _, n in segs]
    want = O.chunk_and_digest_suggested(cfg, data, segs, sugg)
    plain = O.chunk_and_digest(cfg, data, segs)
    assert want.size != plain.size or not np.array_equal(want["end"], plain["end"])  # the suggestions do change cuts
    got = eng.chunk_and_digest(data, segs, suggested=sugg)          # host submit
    assert records_equal(got, want), describe_mismatch(got, want)
    buf = eng.alloc(data.size)
    buf.upload(data)
    got = eng.chunk_and_digest(buf, segs, suggested=sugg)           # device submit
    assert records_equal(got, want), describe_mismatch(got, want)
    # no suggestions given == the plain entry point
    got = eng.chunk_and_digest(buf, segs, suggested=[[] for _ in segs])
    assert records_equal(got, plain)
    buf.free()


def test_upstream_suggested_boundary_vector_on_gpu(engines, O):
    """The chunk sizes recalled from upstream Proxmox's test_suggested_boundary (LE u32 counter buffer, avg 64 KiB):
    [32768, 110609, 229376, 32768, 262144, 262144, 118767] — see tests/test_oracle_buzhash.py for the provenance."""
    eng = engines(65536)
    data = np.arange(256 * 1024, dtype="<u4").view(np.uint8)
    got = eng.chunk_and_digest(data, None, suggested=[[32 * 1024, 32 * 1024, 372753, 405521]])
    assert np.diff(np.concatenate([[0], got["end"]])).astype(int).tolist() == \
        [32768, 110609, 229376, 32768, 262144, 262144, 118767]


def test_stream_position_counts_injected_bytes(engines, O):
    """Encoder().PayloadPosition() after InjectChunks (commit_reuse.go:265, keepLast_chunk_test.go enc.Advance):
    position = written + injected = the coordinate system of the record ends."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    a, b = O.fill(150_000, 5), O.fill(70_001, 6)
    ps = PayloadStream(eng, window_bytes=1 << 17)
    ps.write(a)
    assert ps.position() == a.size and ps.bytes_written() == a.size
    ps.inject(1_000_000)
    assert ps.position() == a.size + 1_000_000 and ps.bytes_written() == a.size
    ps.write(b)
    ps.finish()
    recs = ps.poll()
    assert ps.position() == a.size + 1_000_000 + b.size
    assert int(recs["end"][-1]) == ps.position()
    ps.close()


def test_stream_suggest_matches_the_payload_chunker(engines, O):
    """pbsgpu_stream_suggest at absolute payload positions, windows + carry-over, an inject in the middle."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(77)
    a, b = O.fill(700_003, 11, 3), O.fill(400_000, 12, 0)
    sa = sorted(int(x) for x in rng.integers(1, a.size, 60))
    sb = sorted(int(x) for x in rng.integers(1, b.size, 30))
    inj = 123_456
    ps = PayloadStream(eng, window_bytes=1 << 16)
    pos, si = 0, 0
    while pos < a.size:                       # suggestions are sent when the writer reaches them ("a file starts here")
        nxt = sa[si] if si < len(sa) else a.size
        n = min(int(rng.integers(1, 50_000)), max(nxt - pos, 1), a.size - pos)
        ps.write(a[pos:pos + n])
        pos += n
        while si < len(sa) and sa[si] <= pos:
            ps.suggest()                      # == ps.suggest(pos): "a file starts here"
            si += 1
    ps.inject(inj)
    for x in sb:                              # all boundaries of the second section announced ahead of the bytes
        ps.suggest(a.size + inj + x)
    ps.write(b)
    ps.finish()
    got = ps.poll()
    wa = O.chunk_and_digest_suggested(cfg, a, [(0, a.size)], [sa])
    wb = O.chunk_and_digest_suggested(cfg, b, [(0, b.size)], [sb])
    want_end = np.concatenate([wa["end"], wb["end"] + np.uint64(a.size + inj)])
    assert np.array_equal(got["end"], want_end), (got["end"][:8], want_end[:8])
    assert np.array_equal(got["digest"], np.concatenate([wa["digest"], wb["digest"]]))
    ps.close()


def test_many_streams_and_batches_share_one_engine(engines, O):
    """8 writer threads (one PayloadStream each) + 2 batch threads + helper calls on ONE engine with 2 ticket
    slots: nothing ever answers E_BUSY except a batch submit with both tickets out, and every result is exact."""
    from pbs_plus_amd import PayloadStream, PbsGpuError, _lib

    eng = engines(4096)
    cfg = O.new_config(4096)
    datas = [O.fill(900_000 + 7777 * i, 500 + i, i % 4) for i in range(8)]
    wants = [O.chunk_and_digest(cfg, d) for d in datas]
    errors = []

    def writer(i):
        try:
            for rep in range(3):
                ps = PayloadStream(eng, window_bytes=(1 << 16) << (i % 3))
                rng = np.random.default_rng(i * 10 + rep)
                pos = 0
                while pos < datas[i].size:
                    n = int(rng.integers(1, 120_000))
                    ps.write(datas[i][pos:pos + n])
                    pos += n
                ps.finish()
                got = ps.poll()
                if not (np.array_equal(got["end"], wants[i]["end"]) and np.array_equal(got["digest"], wants[i]["digest"])):
                    errors.append((i, "stream mismatch"))
                ps.close()
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    def batcher(i):
        try:
            for _ in range(6):
                while True:
                    try:
                        got = eng.chunk_and_digest(datas[i])
                        break
                    except PbsGpuError as exc:      # both tickets held by the other batch thread: legal for submit only
                        if exc.status != _lib.E_BUSY:
                            raise
                if not records_equal(got, wants[i]):
                    errors.append((i, "batch mismatch"))
                d = eng.sha256_many(datas[i], [(0, 1000), (5, datas[i].size - 5)])
                import hashlib
                if bytes(d[1]) != hashlib.sha256(datas[i][5:].tobytes()).digest():
                    errors.append((i, "sha256_many mismatch"))
        except Exception as exc:  # noqa: BLE001
            errors.append((i, repr(exc)))

    ts = [threading.Thread(target=writer, args=(i,)) for i in range(8)] + \
         [threading.Thread(target=batcher, args=(i,)) for i in range(2)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=300)
    assert not any(t.is_alive() for t in ts), "a thread hung"
    assert not errors, errors


def test_helper_calls_wait_for_a_context_instead_of_failing(engines, O):
    """with every ticket slot held by uncollected batches, streams, chunkers and the synchronous helpers
    still work (they own / lease their contexts); 8 threads of helper calls queue for the 4 aux contexts."""
    import hashlib

    import xxhash

    from pbs_plus_amd import Chunker, PayloadStream

    eng = engines(4096, 1)
    cfg = O.new_config(4096)
    data = O.fill(600_000, 31, 0)
    held = eng.submit(data)                 # the only ticket slot stays occupied during everything below
    ps = PayloadStream(eng, window_bytes=1 << 17)
    ps.write(data)
    ps.finish()
    want = O.chunk_and_digest(cfg, data)
    got = ps.poll()
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    ps.close()
    ck = Chunker(eng)
    assert ck.scan(data) == int(want["end"][0])
    ck.close()
    errors = []

    def helper(i):
        try:
            for _ in range(5):
                seg = [(i * 100, 50_000 + i)]
                if bytes(eng.sha256_many(data, seg)[0]) != hashlib.sha256(data[i * 100:i * 100 + 50_000 + i].tobytes()).digest():
                    errors.append("sha")
                if int(eng.xxh3_many(data, seg)[0]) != xxhash.xxh3_64_intdigest(data[i * 100:i * 100 + 50_000 + i].tobytes()):
                    errors.append("xxh3")
        except Exception as exc:  # noqa: BLE001
            errors.append(repr(exc))

    ts = [threading.Thread(target=helper, args=(i,)) for i in range(8)]
    for t in ts:
        t.start()
    for t in ts:
        t.join(timeout=120)
    assert not errors, errors
    assert records_equal(eng.collect(held), want)


def test_more_than_4096_segments_in_one_submit(engines, O):
    eng = engines(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(4096)
    lens = rng.integers(0, 40_000, 6000)
    lens[::97] = 0
    segs, off = [], 0
    for n in lens:
        segs.append((off, int(n)))
        off += (int(n) + 7) & ~7
    data = O.fill(off, 8, 3)
    want = O.chunk_and_digest(cfg, data, segs)
    got = eng.chunk_and_digest(data, segs)
    assert records_equal(got, want), describe_mismatch(got, want)


def test_sixteen_tickets_in_flight_collected_out_of_order(engines, O):
    eng = engines(65536, 16)
    cfg = O.new_config(65536)
    n = 3 << 20
    bufs, wants = [], []
    for i in range(16):
        b = eng.alloc(n)
        eng.fill(b.ptr, n, 700 + i, i % 4)
        bufs.append(b)
        wants.append(O.chunk_and_digest(cfg, O.fill(n, 700 + i, i % 4)))
    for rep in range(2):
        tickets = [eng.submit(b, None, n) for b in bufs]
        from pbs_plus_amd import PbsGpuError, _lib
        with pytest.raises(PbsGpuError) as ei:
            eng.submit(bufs[0], None, n)            # a 17th: the documented E_BUSY
        assert ei.value.status == _lib.E_BUSY
        order = list(np.random.default_rng(rep).permutation(16))
        for i in order:
            got = eng.collect(tickets[i])
            assert records_equal(got, wants[i]), (i, describe_mismatch(got, wants[i]))
    for b in bufs:
        b.free()


def test_dense_tiles_compact_in_order_at_the_large_tile_size(engines, O):
    """k_compact's per-tile sort must stay linear on crafted input. 64 MiB of a 64-byte-periodic block
    that hits the break test: ~one candidate per period, thousands per 272 KiB scan tile (counting-sort path)."""
    eng = engines(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(123)
    for _ in range(4000):
        block = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(block, 4)).size:
            break
    else:
        pytest.skip("no dense pattern found")
    data = np.tile(block, (64 << 20) // 64)
    buf = eng.alloc(data.size)
    buf.upload(data)
    got = eng.candidates(buf, data.size)
    want = O.candidates(cfg, data)
    assert want.size >= (1 << 20) - 4
    assert np.array_equal(got, want)
    buf.free()


def test_gather_applies_a_piece_table(engines):
    eng = engines(4096)
    rng = np.random.default_rng(9)
    src = rng.integers(0, 256, 3_000_017, dtype=np.uint8)
    sb = eng.alloc(src.size)
    sb.upload(src)
    items, pos = [], 0
    for _ in range(300):
        ln = int(rng.integers(1, 40_000))
        so = int(rng.integers(0, src.size - ln))
        items.append((so, pos, ln))
        pos += ln
    items.append((5, pos, 2_900_000))      # a piece that is split internally (4 MiB work items) and misaligned
    pos += 2_900_000
    db = eng.alloc(pos + 64)
    rng.shuffle(items)
    eng.gather(sb, db, np.array(items, dtype=np.uint64))
    got = db.download(0, pos)
    want = np.zeros(pos, dtype=np.uint8)
    for so, do, ln in items:
        want[do:do + ln] = src[so:so + ln]
    assert np.array_equal(got, want)
    sb.free()
    db.free()


def test_host_fed_stream_at_production_chunk_size(engines, O):
    """The WriteEntryReader seam at avg 4 MiB: 1.5 GiB of host bytes through 128 MiB windows (carry-over of open
    chunks up to 16 MiB, a dozen windows whose chunks share hash jobs) == the batch path on the same bytes, and a
    sampled prefix == the oracle."""
    from pbs_plus_amd import PayloadStream

    eng = engines(4 << 20)
    n = 1536 << 20
    data = np.empty(n, dtype=np.uint8)
    for i in range(0, n, 256 << 20):                       # zero extents make max-size (16 MiB) chunks
        O.fill(256 << 20, 40 + i, 3 if (i >> 28) % 2 else 0, out=data[i:i + (256 << 20)])
    data[700 << 20:760 << 20] = 0
    ps = PayloadStream(eng, window_bytes=128 << 20)
    for pos in range(0, n, 24 << 20):
        ps.write(data[pos:pos + (24 << 20)])
    ps.finish()
    got = ps.poll()
    ps.close()
    buf = eng.alloc(n)
    buf.upload(data)
    want = eng.chunk_and_digest(buf, None, n)
    buf.free()
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    assert int(got["size"].max()) == 16 << 20
    k = 192 << 20
    pre = O.chunk_and_digest(O.new_config(4 << 20), data[:k])
    m = pre.size - 1
    assert m > 10 and np.array_equal(pre["end"][:m], got["end"][:m]) and np.array_equal(pre["digest"][:m], got["digest"][:m])


def test_engine_may_be_closed_before_its_streams(gpu_lib, O):
    """a Go finalizer / Python __del__ may destroy the engine first; children keep it alive."""
    from pbs_plus_amd import Chunker, Engine, PayloadStream, buzhash

    eng = Engine(buzhash.NewConfig(4096), device=0, inflight=1)
    ps = PayloadStream(eng, window_bytes=1 << 16)
    ck = Chunker(eng)
    data = O.fill(300_000, 3, 0)
    ps.write(data[:100_000])
    eng.close()                      # the handle is gone, the engine lives on in its children
    ps.write(data[100_000:])
    ps.finish()
    got = ps.poll()
    want = O.chunk_and_digest(O.new_config(4096), data)
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    assert ck.scan(data) == int(want["end"][0])
    ps.close()
    ck.close()


def test_xxh3_wave_kernel_all_length_classes_and_alignments(engines):
    """k_xxh3 (one wave per input): every length class of XXH3-64 (0, 1-3, 4-8, 9-16, 17-128, 129-240, the 1 KiB block
    edges, multi-block) at every byte alignment, against the xxhash C library."""
    import xxhash

    eng = engines(4096)
    rng = np.random.default_rng(31)
    data = rng.integers(0, 256, 6_000_000, dtype=np.uint8)
    lens = [0, 1, 2, 3, 4, 7, 8, 9, 16, 17, 31, 32, 33, 64, 65, 96, 97, 128, 129, 130, 161, 192, 239, 240, 241, 255, 256,
            257, 511, 512, 513, 1023, 1024, 1025, 1087, 1088, 1089, 2047, 2048, 2049, 3000, 4096, 10_000, 65_536 + 5,
            1_000_003, 3_145_728]
    segs, off = [], 0
    for i, n in enumerate(lens):
        o = off + (i % 8)                      # rotate through all 8 byte alignments
        segs.append((o, n))
        off = (o + n + 15) & ~7
    assert off <= data.size
    buf = eng.alloc(data.size)
    buf.upload(data)
    got = eng.xxh3_many(buf, segs)
    want = [xxhash.xxh3_64_intdigest(data[o:o + n].tobytes()) for o, n in segs]
    bad = [(n, hex(int(g)), hex(w)) for (o, n), g, w in zip(segs, got, want) if int(g) != w]
    assert not bad, bad[:5]
    buf.free()


def test_stream_file_tee_matches_xxhash_and_payload_layout(engines, O):
    """A4: the per-file XXH3-64 tee of the stream writer + pxar payload entries. Files of every awkward size (0, tiny,
    around the 240 / 1024 byte edges, larger than a window so that one file spans several windows, files ending exactly
    on a window edge), written with random write sizes through small windows: every hash == xxhash of the file body,
    the stream bytes == the .ppxar layout (start marker, {16-byte header, body}*, tail marker) and its cuts/digests ==
    the oracle on that layout."""
    import struct

    import xxhash

    from pbs_plus_amd import PayloadStream, _lib

    eng = engines(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(5)
    window = 1 << 16
    sizes = [0, 1, 100, 240, 241, 1023, 1024, 1025, 5000, window - 16, window, 3 * window + 77, 17, 0, 200_000, 1088,
             window * 2, 9]
    fmt = _lib.PayloadFormat()
    _lib.check(_lib.lib().pbsgpu_payload_format_default(fmt), "fmt")
    ps = PayloadStream(eng, window_bytes=window)
    expect = bytearray()
    ps.write_marker(False)
    expect += struct.pack("<QQ", fmt.start_type, 16)
    bodies, offsets, idxs = [], [], []
    for k, n in enumerate(sizes):
        body = rng.integers(0, 256, n, dtype=np.uint8) if k % 5 else np.zeros(n, dtype=np.uint8)
        off = ps.begin_entry(n)
        assert off == len(expect)
        expect += struct.pack("<QQ", fmt.payload_type, 16 + n)
        pos = 0
        while pos < n:
            m = min(int(rng.integers(1, 50_000)), n - pos)
            if k % 2:
                ps.write(body[pos:pos + m])
            else:                                 # zero-copy feed
                r = ps.reserve()
                m = min(m, r.size)
                r[:m] = body[pos:pos + m]
                ps.commit(m)
            pos += m
        idxs.append(ps.end_entry())
        expect += body.tobytes()
        bodies.append(body)
        offsets.append(off)
    ps.write_marker(True)
    expect += struct.pack("<QQ", fmt.tail_type, 16)
    ps.finish()
    files = ps.poll_files()
    assert [f[0] for f in files] == idxs == list(range(len(sizes)))
    assert [f[1] for f in files] == sizes
    bad = [(i, n) for (i, n, h), b in zip(files, bodies) if h != xxhash.xxh3_64_intdigest(b.tobytes())]
    assert not bad, bad
    got = ps.poll()
    want = O.chunk_and_digest(cfg, np.frombuffer(bytes(expect), dtype=np.uint8))
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    assert ps.position() == len(expect)
    # a short body is the Go writer's "unexpected EOF"
    ps2 = PayloadStream(eng, window_bytes=window)
    ps2.begin_entry(10)
    ps2.write(np.zeros(4, np.uint8))
    with pytest.raises(_lib.PbsGpuError):
        ps2.end_entry()
    with pytest.raises(_lib.PbsGpuError):
        ps2.write(np.zeros(7, np.uint8))      # more than the header announced
    ps2.close()
    ps.close()


def test_stream_tee_large_files_at_production_window(engines, O):
    """The tee at production sizes: avg 4 MiB, 64 MiB windows, a 200 MiB file followed by a run of small ones."""
    import xxhash

    from pbs_plus_amd import PayloadStream

    eng = engines(4 << 20)
    big = O.fill(200 << 20, 3, 0)
    small = [O.fill(n, 50 + i, 0) for i, n in enumerate([4096, 1 << 20, 333, 7 << 20])]
    ps = PayloadStream(eng, window_bytes=64 << 20)
    for body in [big] + small:
        ps.begin_file()
        for pos in range(0, body.size, 24 << 20):
            ps.write(body[pos:pos + (24 << 20)])
        ps.end_file()
    ps.finish()
    files = ps.poll_files()
    assert [f[2] for f in files] == [xxhash.xxh3_64_intdigest(b.tobytes()) for b in [big] + small]
    got = ps.poll()
    want = O.chunk_and_digest(O.new_config(4 << 20), np.concatenate([big] + small))
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    ps.close()


@pytest.mark.parametrize("workload,extra", [("stream64g", ["--slots", "2"]), ("corpus_dup", ["--file-mib", "8", "--scaling", "strong"]),
                                             ("corpus_dup", ["--file-mib", "8"]),
                                             # two rings on ONE GPU: each service gets 64 CUs (a real job has one rank per GPU)
                                             ("ring", ["--ring-streams", "2", "--arena-gib", "1.5", "--ring-sha-cus", "64"])])
def test_bench_two_ranks_share_the_gpu(gpu_lib, workload, extra):
    """bench.py's N > 1 branch with the REAL engine: two ranks over gloo on the one GPU of the test box (the driver
    runs the same code with RCCL, one rank per GPU). Rank 0 prints the aggregate line; the digest-set reduce ran."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    cmd = [sys.executable, os.path.join(root, "bench.py"), "--gpus", "2", "--workload", workload, "--gib", "0.25",
           "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--avg", str(1 << 20)] + extra
    procs = []
    for rank in range(2):
        env = dict(os.environ, RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE="2", MASTER_ADDR="127.0.0.1",
                   MASTER_PORT=str(port), PBS_BENCH_BACKEND="gloo")
        procs.append(subprocess.Popen(cmd, env=env, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True))
    outs = [p.communicate(timeout=300) for p in procs]
    assert all(p.returncode == 0 for p in procs), outs
    lines = [[ln for ln in o.splitlines() if ln.startswith("{")] for o, _ in outs]
    assert len(lines[0]) == 1 and not lines[1]
    d = json.loads(lines[0][0])
    assert d["n_gpus"] == 2 and d["value"] > 0
    if workload == "corpus_dup":
        dd = d["results"]["dedup"]
        assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9 and dd["expected_duplicate_frac"] > 0.2
    else:
        assert d["results"]["dedup_last_step"]["nrecords"] > 0


@pytest.mark.parametrize("seed", range(int(__import__("os").environ.get("PBS_SOAK_SEEDS", "24"))))
def test_stream_random_programs_match_the_model(engines, O, seed):
    """Randomised programs over the whole stream surface — write / reserve+commit with random sizes, suggested
    boundaries announced ahead of the data, InjectChunks cuts, file tees, payload entries, polls at random points,
    window sizes from 16 KiB up — checked against a host model built from the oracle (payload chunker per section)
    and the xxhash library. Exercises the deferred-cut / headroom-carry / shared-hash-job machinery at every edge."""
    import struct

    import xxhash

    from pbs_plus_amd import PayloadStream, _lib

    avg = [256, 4096, 4096, 65536][seed % 4]
    eng = engines(avg)
    cfg = O.new_config(avg)
    rng = np.random.default_rng(1000 + seed)
    window = int(rng.choice([cfg.max, cfg.max * 2, cfg.max * 4 + 4096]))
    fmt = _lib.PayloadFormat()
    _lib.check(_lib.lib().pbsgpu_payload_format_default(fmt), "fmt")
    ps = PayloadStream(eng, window_bytes=window)
    sections = [bytearray()]          # bytes of each section (between injects)
    sec_start = [0]                   # absolute payload position of each section's first byte
    sugg = [[]]                       # suggested boundaries per section, relative
    files, got_recs, got_files = [], [], []
    pos = 0                           # absolute payload position (written + injected)
    last_sugg = 0

    def feed(data):
        nonlocal pos
        data = np.frombuffer(bytes(data), dtype=np.uint8)
        o = 0
        while o < data.size:
            n = min(int(rng.integers(1, 3 * cfg.max)), data.size - o)
            if rng.random() < 0.5:
                ps.write(data[o:o + n])
            else:
                r = ps.reserve()
                n = min(n, r.size)
                r[:n] = data[o:o + n]
                ps.commit(n)
            o += n
            if rng.random() < 0.2:
                got_recs.append(ps.poll())
            if rng.random() < 0.1:
                got_files.extend(ps.poll_files())
        sections[-1] += bytes(data)
        pos += data.size

    for _ in range(int(rng.integers(4, 14))):
        op = rng.random()
        body = O.fill(int(rng.integers(0, 6 * cfg.max)), int(rng.integers(1, 1 << 30)), int(rng.integers(0, 4))) \
            if rng.random() < 0.9 else np.zeros(int(rng.integers(0, 300)), dtype=np.uint8)
        if op < 0.15 and len(sections[-1]) >= 0:
            inj = int(rng.integers(0, 5 * cfg.max))
            ps.inject(inj)
            pos += inj
            sections.append(bytearray())
            sec_start.append(pos)
            sugg.append([])
            continue
        # boundaries somewhere in the bytes about to be written (announced now, i.e. ahead of the data)
        span = body.size + (16 if op >= 0.6 else 0)
        for b in sorted(int(x) for x in rng.integers(0, max(span, 1) + 1, int(rng.integers(0, 4)))):
            a = pos + b
            if a >= last_sugg:
                ps.suggest(a)
                last_sugg = a
                sugg[-1].append(a - sec_start[-1])
        if op < 0.6:
            tee = rng.random() < 0.5
            if tee:
                ps.begin_file()
            feed(body)
            if tee:
                files.append((ps.end_file(), body.size, xxhash.xxh3_64_intdigest(body.tobytes())))
        else:
            off = ps.begin_entry(body.size)
            assert off == pos
            sections[-1] += struct.pack("<QQ", fmt.payload_type, 16 + body.size)
            pos += 16
            feed(body)
            files.append((ps.end_entry(), body.size, xxhash.xxh3_64_intdigest(body.tobytes())))
    assert ps.position() == pos
    ps.finish()
    got_recs.append(ps.poll())
    got_files.extend(ps.poll_files())
    got = np.concatenate(got_recs)
    want_end, want_dig, want_sec = [], [], []
    for k, (sec, st, sg) in enumerate(zip(sections, sec_start, sugg)):
        if not sec:
            continue
        w = O.chunk_and_digest_suggested(cfg, np.frombuffer(bytes(sec), dtype=np.uint8), [(0, len(sec))], [sorted(sg)])
        want_end.append(w["end"] + np.uint64(st))
        want_dig.append(w["digest"])
        want_sec += [k] * w.size
    if want_end:
        assert np.array_equal(got["end"], np.concatenate(want_end)), (seed, got["end"][:6], np.concatenate(want_end)[:6])
        assert np.array_equal(got["digest"], np.concatenate(want_dig)), seed
        assert got["segment"].tolist() == want_sec
    else:
        assert got.size == 0
    assert got_files == files, (seed, got_files[:3], files[:3])
    ps.close()


@pytest.mark.parametrize("workload", ["manyfiles", "corpus_dup", "rechunk"])
def test_bench_workloads_at_full_scale_check_against_the_oracle(gpu_lib, workload):
    """BASELINE configs[2..4] at their full single-GPU shapes (2 x 128 GiB of 64 MiB files / a 128 GiB corpus share
    with 40 % duplicated segments / the share after 2 % edits): bench.py's sampled oracle check must hold on the
    records of a full-size pass, and the workload-level result must be the planted one."""
    import json
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--workload", workload, "--steps", "4", "--warmup", "0",
                          "--cpu-sample-gib", "0.5"], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, out.stderr[-2000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["cpu_baseline"]["records_match_gpu"] is True and d["cpu_baseline"]["records_checked"] > 50
    sp = d["cpu_baseline"]["whole_batch_restart_points"]   # every resident slot, spread over its bytes
    assert sp["ok"] is True and sp["slots"] == d["config"]["resident_batches"] and sp["records_checked"] > 300, sp
    assert d["config"]["resident_bytes_per_gpu"] >= 120 * (1 << 30)
    if workload == "manyfiles":
        assert d["config"]["files_per_batch"] == 2048 and d["config"]["chunks_per_batch"] > 20_000
    if workload == "corpus_dup":
        dd = d["results"]["dedup"]
        assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9 and 0.3 < dd["expected_duplicate_frac"] < 0.5
    if workload == "rechunk":
        assert 0.5 < d["results"]["reused_chunk_bytes_frac"] < 0.95


def test_cpp_mirror_payload_entries_and_tee(gpu_lib, O, tmp_path):
    """include/pbsgpu.hpp, round-2 surface: WritePayloadEntry (header + body + XXH3 tee), WriteMarker, InjectChunks,
    SuggestBoundary, BackedHashes — a C++ program written against the mirror, checked against xxhash and the oracle
    run on the layout the program is supposed to have produced."""
    import os
    import struct
    import subprocess

    import xxhash

    from pbs_plus_amd import _lib

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "test_cpp_tee")
    libdir = os.path.join(root, "pbs_plus_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(root, "tests", "native", "test_cpp_tee.cpp"),
                    "-L" + libdir, "-lpbsgpu", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpp-tee-ok" in out.stdout, out.stdout + out.stderr
    fmt = _lib.PayloadFormat()
    _lib.check(_lib.lib().pbsgpu_payload_format_default(fmt), "fmt")
    sizes = [300000, 0, 17, 70000, 1024, 200001]
    bodies = [O.fill(n, 100 + k, 0) for k, n in enumerate(sizes)]
    inj = 123456
    sec0 = struct.pack("<QQ", fmt.start_type, 16)
    sec1, offs, pos = b"", [], 16
    sugg1 = []
    for k, b in enumerate(bodies):
        hdr = struct.pack("<QQ", fmt.payload_type, 16 + b.size)
        if k <= 1:
            offs.append(len(sec0))
            sec0 += hdr + b.tobytes()
        else:
            if k == 3:
                sugg1.append(len(sec1))
            offs.append(len(sec0) + inj + len(sec1))
            sec1 += hdr + b.tobytes()
    sec1 += struct.pack("<QQ", fmt.tail_type, 16)
    cfg = O.new_config(4096)
    w0 = O.chunk_and_digest(cfg, np.frombuffer(sec0, dtype=np.uint8))
    w1 = O.chunk_and_digest_suggested(cfg, np.frombuffer(sec1, dtype=np.uint8), [(0, len(sec1))], [sugg1])
    want = [(int(r["end"]), int(r["size"]), bytes(r["digest"]).hex()) for r in w0] + \
           [(int(r["end"]) + len(sec0) + inj, int(r["size"]), bytes(r["digest"]).hex()) for r in w1]
    flines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("F ")]
    clines = [ln.split() for ln in out.stdout.splitlines() if ln.startswith("C ")]
    assert [(int(f[1]), int(f[2]), int(f[3], 16), int(f[4])) for f in flines] == \
        [(k, sizes[k], xxhash.xxh3_64_intdigest(bodies[k].tobytes()), offs[k]) for k in range(6)]
    assert [(int(c[1]), int(c[2]), c[3]) for c in clines] == want
    assert int(out.stdout.split("cpp-tee-ok")[1]) == len(sec0) + inj + len(sec1)


def test_parallel_resolve_equals_the_serial_walk_on_awkward_streams(gpu_lib):
    """k_resolve_par (pointer doubling over the candidate list) on everything the serial walk handles: small averages
    with thousands of candidates, zero runs (forced max-size cuts only), a candidate exactly at the stream end, streams
    shorter than min, dense periodic candidates (falls back when they exceed the node capacity), min = 64. Forced on for
    every size through PBSGPU_RESOLVE_PAR_MIN=0 in a subprocess (the switch is read once per process)."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from pbs_plus_amd import Engine, buzhash\n"
        "from tests.helpers import records_equal, describe_mismatch\n"
        "rng = np.random.default_rng(7)\n"
        "for avg in (256, 4096, 65536, 4 << 20):\n"
        "    eng = Engine(buzhash.NewConfig(avg)); cfg = O.new_config(avg)\n"
        "    cases = [O.fill(avg * 300 + 17, 5, 0), O.fill(avg * 200, 6, 3), np.zeros(avg * 40 + 5, np.uint8), O.fill(63, 7, 0),\n"
        "             O.fill(cfg.min - 1, 8, 0), O.fill(cfg.max, 9, 1), O.fill(1, 10, 0), O.fill(avg * 64, 11, 2)]\n"
        "    if avg == 256:\n"
        "        cases += [O.fill(64 << 20, 13, 0), O.fill(48 << 20, 14, 3)]   # ~400 k candidates: the grid-wide doubling variant\n"
        "    if avg == 4096:\n"
        "        for _ in range(4000):\n"
        "            block = rng.integers(0, 256, 64, dtype=np.uint8)\n"
        "            if O.candidates(cfg, np.tile(block, 4)).size: break\n"
        "        cases += [np.tile(block, 3000), np.tile(block, 600_000)]   # dense: 3 k candidates / 600 k (> node capacity -> serial)\n"
        "        d = O.fill(avg * 50, 12, 0); c = O.candidates(cfg, d)\n"
        "        cases.append(d[:int(c[len(c) // 2])])                      # a candidate exactly at the stream end\n"
        "    for i, data in enumerate(cases):\n"
        "        got, want = eng.chunk_and_digest(data), O.chunk_and_digest(cfg, data)\n"
        "        assert records_equal(got, want), (avg, i, describe_mismatch(got, want))\n"
        "    eng.close()\n"
        "print('par-ok')\n" % root)
    env = dict(os.environ, PBSGPU_RESOLVE_PAR_MIN="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
    assert "par-ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]


def test_dense_sha_form_stays_bit_exact(gpu_lib):
    """The 4-pairs-per-CU form of the SHA-256 kernel (k_sha256_pair<..., true>) is chosen per batch when the work exceeds
    PBSGPU_SHA_DENSE_PCT % of the longest chain per lane. PBSGPU_SHA_DENSE_PCT=1 (read once per process -> subprocess)
    makes it the form of every launch that has >~330 longest-chunks' worth of blocks: batch records (decided from the
    batch's bytes and the chunker's maximum), whole-segment hashing with every padding length and misaligned starts, and
    the stream writer's shared hash jobs (decided from the items of the launch) — all against the oracle / hashlib."""
    import os
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = (
        "import hashlib, numpy as np, sys\n"
        "sys.path.insert(0, %r)\n"
        "from oracle import oracle as O\n"
        "from pbs_plus_amd import Engine, buzhash\n"
        "from pbs_plus_amd.engine import PayloadStream\n"
        "from tests.helpers import records_equal, describe_mismatch\n"
        "for avg, n, kind in ((4096, 40_000_001, 0), (4096, 24_000_000, 3), (65536, 300_000_017, 0)):\n"
        "    eng = Engine(buzhash.NewConfig(avg)); cfg = O.new_config(avg)\n"
        "    data = O.fill(n, 21, kind)\n"
        "    got, want = eng.chunk_and_digest(data), O.chunk_and_digest(cfg, data)\n"
        "    assert records_equal(got, want), (avg, describe_mismatch(got, want))\n"
        "    if avg == 4096:\n"
        "        ps = PayloadStream(eng, window_bytes=32 << 20)\n"
        "        for off in range(0, n, 7_000_003): ps.write(data[off:off + 7_000_003])\n"
        "        ps.finish(); recs = ps.poll(); ps.close()\n"
        "        assert records_equal(recs, want), ('stream', describe_mismatch(recs, want))\n"
        "    eng.close()\n"
        "eng = Engine(buzhash.NewConfig(4096))\n"
        "blob = O.fill(6_000_000, 22, 0)\n"
        "rng = np.random.default_rng(3)\n"
        "segs = [(int(o), int(l)) for o, l in zip(rng.integers(0, 5_000_000, 30000), rng.integers(0, 200, 30000))]\n"
        "segs += [(i, l) for i in range(4) for l in range(0, 131)]   # longest item 3 blocks: dense by a wide margin\n"
        "dig = eng.sha256_many(blob, segs)\n"
        "for i in list(range(0, 30000, 97)) + list(range(30000, len(segs))):\n"
        "    o, l = segs[i]\n"
        "    assert bytes(dig[i]) == hashlib.sha256(blob[o:o + l].tobytes()).digest(), (i, o, l)\n"
        "eng.close()\n"
        "print('dense-ok')\n" % root)
    for extra in ({},):
        env = dict(os.environ, PBSGPU_SHA_DENSE_PCT="1", **extra)
        out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
        assert "dense-ok" in out.stdout, str(extra) + out.stdout[-2000:] + out.stderr[-3000:]


@pytest.mark.parametrize("workload", ["stream64g", "corpus_dup", "ring"])
def test_bench_rccl_code_path_with_one_rank(gpu_lib, workload):
    """The "nccl" (= RCCL) branch of bench.py on real hardware: process-group init with device_id, barrier, all_reduce
    (MAX / SUM), all_gather_into_tensor of the uint8 record payload, destroy — forced on with ONE rank
    (PBS_BENCH_FORCE_DIST), because the test box has a single GPU. The 8-GPU runs of the driver take exactly this path."""
    import json
    import os
    import socket
    import subprocess
    import sys

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with socket.socket() as so:
        so.bind(("127.0.0.1", 0))
        port = so.getsockname()[1]
    env = dict(os.environ, RANK="0", LOCAL_RANK="0", WORLD_SIZE="1", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port),
               PBS_BENCH_FORCE_DIST="1", PBS_BENCH_BACKEND="nccl")
    extra = {"stream64g": ["--slots", "2"], "corpus_dup": ["--file-mib", "8"],
             "ring": ["--ring-streams", "2", "--arena-gib", "2"]}[workload]
    out = subprocess.run([sys.executable, os.path.join(root, "bench.py"), "--gpus", "1", "--workload", workload, "--gib", "0.25",
                          "--steps", "4", "--warmup", "2", "--no-cpu-baseline", "--avg", str(1 << 20)] + extra,
                         env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    assert d["n_gpus"] == 1 and d["value"] > 0
    if workload == "corpus_dup":
        dd = d["results"]["dedup"]
        assert abs(dd["duplicate_bytes_frac"] - dd["expected_duplicate_frac"]) < 1e-9
    else:
        assert d["results"]["dedup_last_step"]["nrecords"] > 0
