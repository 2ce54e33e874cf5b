"""Round-3 GPU tests (through the C ABI): the reader-buffer rule of suggested boundaries (feed size / grid), the digest-set
reduce on records that already are in device memory, the whole-file hash policy query and the window-pool trim."""
import ctypes as C

import numpy as np
import pytest

from helpers import describe_mismatch, records_equal

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def eng4k(gpu_lib):
    from pbs_plus_amd import Engine, buzhash

    e = Engine(buzhash.NewConfig(4096), device=0, inflight=2)
    yield e
    e.close()


def _ends_to_records(O, cfg, data, ends):
    """(end, size, digest) records of one stream from its chunk END offsets (hashlib-independent: oracle SHA-256)"""
    recs = np.zeros(len(ends), dtype=O.RECORD_DTYPE)
    start = 0
    for i, e in enumerate(ends):
        e = int(e)
        recs[i]["end"], recs[i]["size"], recs[i]["segment"] = e, e - start, 0
        recs[i]["digest"] = np.frombuffer(O.sha256(data[start:e]), dtype=np.uint8)
        start = e
    return recs


@pytest.mark.parametrize("feed,absolute", [(1, False), (4096, False), (65536, False), (0, False), (4096, True), (65536, True),
                                           (1000, True), (1 << 20, True)])
def test_suggested_boundaries_follow_the_declared_reader_buffer(eng4k, O, feed, absolute):
    """pbsgpu_engine_set_suggested_feed: the cut list equals the payload chunker's when it is handed `feed` bytes per scan
    call — a boundary inside the current buffer pre-empts an EARLIER hash cut of that buffer (upstream semantics), a hash
    cut of an earlier buffer wins — with the buffer grid restarting at every cut, or anchored at the stream start."""
    cfg = O.new_config(4096)
    rng = np.random.default_rng(1234)
    n = 3_000_017
    data = O.fill(n, 77, 0)
    data[500_000:700_000] = 0                                  # a zero run: max cuts unless a boundary intervenes
    plain = O.chunk_stream(cfg, data)
    # boundaries placed shortly BEHIND hash cuts (the case where the feed size decides) plus random ones
    near = [int(e) + int(d) for e, d in zip(plain[5:400:7], rng.integers(1, 3000, 400))]
    sugg = sorted(set(near + [int(x) for x in rng.integers(1, n, 150)] + [n, 64, 1024, 1025]))
    want_ends = O.chunk_stream_suggested(cfg, data, sugg, feed=feed, absolute=absolute)
    serial_ends = O.chunk_stream_suggested(cfg, data, sugg, feed=1)
    if feed not in (1,):
        assert not np.array_equal(want_ends, serial_ends), "the test input must make the feed size matter"
    eng4k.set_suggested_feed(feed, absolute)
    try:
        got = eng4k.chunk_and_digest(data, [(0, n)], suggested=[sugg])
    finally:
        eng4k.set_suggested_feed(1, False)
    want = _ends_to_records(O, cfg, data, want_ends)
    assert records_equal(got, want), describe_mismatch(got, want)


def test_suggested_feed_through_the_stream_writer(eng4k, O):
    """the same rule in pbsgpu_stream_*: windows, carry-over of the open chunk, absolute grid = payload positions"""
    from pbs_plus_amd import PayloadStream

    cfg = O.new_config(4096)
    rng = np.random.default_rng(99)
    n = 1_200_011
    data = O.fill(n, 78, 0)
    plain = O.chunk_stream(cfg, data)
    sugg = sorted(set([int(e) + int(d) for e, d in zip(plain[3:200:5], rng.integers(1, 2500, 200))] +
                      [int(x) for x in rng.integers(1, n, 60)]))
    for feed, absolute in ((8192, True), (8192, False)):
        want_ends = O.chunk_stream_suggested(cfg, data, sugg, feed=feed, absolute=absolute)
        eng4k.set_suggested_feed(feed, absolute)
        try:
            ps = PayloadStream(eng4k, window_bytes=1 << 16)
            for x in sugg:
                ps.suggest(x)
            pos = 0
            while pos < n:
                m = min(int(rng.integers(1, 40_000)), n - pos)
                ps.write(data[pos:pos + m])
                pos += m
            ps.finish()
            got = ps.poll()
            ps.close()
        finally:
            eng4k.set_suggested_feed(1, False)
        assert np.array_equal(got["end"], want_ends), (feed, absolute, got["end"][:6], want_ends[:6])


def test_dedup_on_device_resident_records(eng4k, O):
    """pbsgpu_dedup_device == pbsgpu_dedup_host on the same set (the RCCL all-gather's receive buffer never goes back to
    the host in dist.global_dedup)."""
    from pbs_plus_amd import RECORD_DTYPE

    rng = np.random.default_rng(5)
    n = 20_000
    recs = np.zeros(n, dtype=RECORD_DTYPE)
    ids = rng.integers(0, 6000, n)
    for i in range(n):
        recs[i]["digest"] = np.frombuffer(int(ids[i]).to_bytes(8, "big") * 4, dtype=np.uint8)
        recs[i]["size"] = 1000 + (int(ids[i]) % 7)
    dup_h, st_h = eng4k.dedup(recs)
    buf = eng4k.alloc(recs.nbytes)
    buf.upload(recs.view(np.uint8).reshape(-1))
    dup_d, st_d = eng4k.dedup_device(buf.ptr, n)
    buf.free()
    assert st_d == st_h and np.array_equal(dup_d, dup_h)
    assert st_h["nunique"] == len(set(ids.tolist())) and int(dup_h.sum()) == n - st_h["nunique"]


def test_sha256_many_policy_and_trim(eng4k):
    L = eng4k._L
    pays = C.c_int(-1)
    assert L.pbsgpu_sha256_many_pays(eng4k._h, 4, 16, C.byref(pays)) == 0 and pays.value == 0        # the reference's 4-file job
    assert L.pbsgpu_sha256_many_pays(eng4k._h, 100_000, 16, C.byref(pays)) == 0 and pays.value == 1
    assert L.pbsgpu_sha256_many_pays(eng4k._h, 56, 1, C.byref(pays)) == 0 and pays.value == 1
    from pbs_plus_amd import PayloadStream
    ps = PayloadStream(eng4k, window_bytes=1 << 20)
    ps.write(np.zeros(100_000, dtype=np.uint8))
    ps.finish()
    ps.poll()
    ps.close()                                   # its window buffers went to the engine's pool
    freed = C.c_uint64()
    assert L.pbsgpu_engine_trim(eng4k._h, C.byref(freed)) == 0 and freed.value > 0
    assert L.pbsgpu_engine_trim(eng4k._h, C.byref(freed)) == 0 and freed.value == 0


def test_archives_back_to_back_finish_begin_and_recycled_stream_contexts(eng4k, O):
    """pbsgpu_stream_finish_begin / pbsgpu_stream_done: a writer closes an archive's input and goes on with the next
    archive while the first one's last chunks are hashed; closed streams hand their contexts (cut contexts, staging, tee
    buffers) to the engine's next stream of the same window size — five generations through the same contexts, with the
    XXH3 tee and a forced cut in some of them, every archive bit-exact vs the oracle."""
    import time

    import xxhash

    from pbs_plus_amd import PayloadStream, _lib

    cfg = O.new_config(4096)
    rng = np.random.default_rng(321)
    draining = []
    checked = 0

    def settle(block):
        nonlocal checked
        for item in list(draining):
            ps, data, want_files, parts = item
            if block:
                ps.finish()
            parts.append(ps.poll())
            if block or ps.done():
                assert ps.done()
                parts.append(ps.poll())
                got = np.concatenate(parts)
                want = O.chunk_and_digest(cfg, data, [(0, data.size)])
                assert records_equal(got, want), describe_mismatch(got, want)
                files = ps.poll_files()
                assert [(f[1], f[2]) for f in files] == want_files
                ps.close()
                draining.remove(item)
                checked += 1

    for gen in range(5):
        n = int(rng.integers(700_000, 2_500_000))
        data = O.fill(n, 500 + gen, gen % 4)
        ps = PayloadStream(eng4k, window_bytes=1 << 20)
        want_files = []
        pos = 0
        while pos < n:
            m = min(int(rng.integers(1, 300_000)), n - pos)
            tee = gen % 2 == 1
            if tee:
                ps.begin_file()
            ps.write(data[pos:pos + m])
            if tee:
                ps.end_file()
                want_files.append((m, xxhash.xxh3_64_intdigest(data[pos:pos + m].tobytes())))
            pos += m
            settle(False)
        ps.finish_begin()
        assert ps.bytes_written() == n
        with pytest.raises(_lib.PbsGpuError) as ei:
            ps.write(data[:10])
        assert ei.value.status == _lib.E_STATE
        draining.append((ps, data, want_files, []))
        assert len(draining) <= 5
    t0 = time.time()
    while draining and time.time() - t0 < 30:
        settle(False)
        time.sleep(0.002)
    assert not draining, "pbsgpu_stream_done never reported the end"
    settle(True)
    assert checked == 5
