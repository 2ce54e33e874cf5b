"""Round 5: the page ring's corner cases the round-4 review found, each against the CPU oracle through the C ABI.

* a FAILED stream that is closed while rounds still carry its segments must not poison the stream that takes its slot next;
* pbsgpu_ring_quiesce must hash rounds that were cut AHEAD of the service (a lone bulk stream on an idle ring);
* a stream may be longer than 1 TiB (52-bit logical offsets; rounds 3-4 stopped at 2^40 with a sticky E_INVALID) — the case of a
  payload stream without a forced cut: a fresh multi-TiB backup, a tape conversion (internal/tapeio/converter.go:827-842);
* device memory freed while a service runs is parked only up to a cap: beyond it the rings park their services, the memory is
  really freed and the services start again — the stream that was being hashed meanwhile stays bit-exact;
* the digest-set reduce reports bad arguments COLLECTIVELY (every rank the same error, nobody left inside the all-gather).
"""
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GiB = 1 << 30


@pytest.fixture(autouse=True)
def _short_idle_timeout(monkeypatch):
    monkeypatch.setenv("PBSGPU_RING_IDLE_TIMEOUT_S", "8")


def _engine(avg):
    from pbs_plus_amd import Engine, buzhash

    return Engine(buzhash.NewConfig(avg), device=0, inflight=1)


def _assert_same(got, want, what):
    assert got.size == want.size, (what, got.size, want.size)
    assert np.array_equal(got["end"], want["end"]), what
    assert np.array_equal(got["size"], want["size"]), what
    assert np.array_equal(got["digest"], want["digest"]), what


def _dense_pattern(O, cfg):
    rng = np.random.default_rng(5)
    for _ in range(20000):  # a 64-byte pattern whose (periodic) window hash passes the break test: one candidate per period
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(p, 8)).size >= 6:
            return p
    return None


@pytest.mark.parametrize("attempt", range(4))
def test_a_failed_stream_closed_early_cannot_poison_the_next_stream_of_its_slot(gpu_lib, O, attempt):
    """The crafted stream commits MANY dense pages at once, so that when the host first learns of the failure (a reaped
    round) two more rounds carrying segments of it are still in flight. The caller closes it at that moment and opens new
    streams until one lands in the same slot (or the ring says busy), feeding ordinary bytes. Round 4 applied the in-flight
    rounds' `failed` / `final` / cells to whatever stream held the slot index by then: the innocent stream was failed with
    E_DENSITY or ended early. Now the slot stays taken until those rounds have been reaped."""
    from pbs_plus_amd import PageRing, PbsGpuError, _lib

    avg = 4096
    eng = _engine(avg)
    cfg = O.new_config(avg)
    pat = _dense_pattern(O, cfg)
    if pat is None:
        pytest.skip("no dense pattern found")
    PAGE = 65536
    ring = PageRing(eng, arena_bytes=64 * (PAGE + 256), page_bytes=PAGE, max_streams=2, sha_cus=4, round_pages=3)
    L = eng._L
    bad = np.concatenate([O.fill(PAGE, 9 + attempt, 0), np.tile(pat, 24 * PAGE // 64)])
    sid = ring.open()
    other = ring.open()                                      # an ordinary stream sharing the rounds (keeps slot 1 taken)
    odata = O.fill(9 * PAGE + 77, 300 + attempt, 0)
    off = ooff = 0
    failed_at = None
    ogot = []
    t0 = time.time()
    while failed_at is None and time.time() - t0 < 30:
        try:
            while off < bad.size:                            # as many pages as the arena gives: several rounds' worth
                r = ring.reserve(sid)
                if r is None:
                    break
                assert L.pbsgpu_memcpy_h2d(eng._h, r[0], bad[off:off + PAGE].ctypes.data, PAGE) == 0
                off += PAGE
                ring.commit(sid, PAGE, final=(off == bad.size))
            if ooff < odata.size:
                r = ring.reserve(other)
                if r is not None:
                    n = min(PAGE, odata.size - ooff)
                    assert L.pbsgpu_memcpy_h2d(eng._h, r[0], odata[ooff:ooff + n].ctypes.data, n) == 0
                    ooff += n
                    ring.commit(other, n, final=(ooff == odata.size))
            ring.pump()
            ring.poll(sid)
            ogot.append(ring.poll(other)[0].copy())
        except PbsGpuError as exc:
            assert exc.status == _lib.E_DENSITY, exc
            failed_at = time.time()
    assert failed_at is not None, "the crafted stream never failed"
    with pytest.raises(PbsGpuError) as ei:
        ring.close_stream(sid)                               # at once: rounds with its segments may still be in flight
    assert ei.value.status == _lib.E_DENSITY
    # the closed stream's id answers nothing any more (until a new stream owns the slot)
    assert L.pbsgpu_ring_commit(ring._h, sid, 0, 1) in (_lib.E_INVALID,)
    # new streams, one after the other, as fast as the ring hands out a slot: every one must be bit-exact
    for k in range(3):
        data = O.fill(5 * PAGE + 1000 * k + 13, 500 + 10 * attempt + k, k % 4)
        nsid = None
        t1 = time.time()
        while nsid is None and time.time() - t1 < 30:
            try:
                nsid = ring.open()
            except PbsGpuError as exc:
                assert exc.status == _lib.E_BUSY, exc        # the dead stream's slot is still held by its last rounds
                ring.pump()
                ogot.append(ring.poll(other)[0].copy())
        assert nsid is not None
        noff, got, fin = 0, [], False
        t1 = time.time()
        while not fin and time.time() - t1 < 30:
            if noff < data.size:
                r = ring.reserve(nsid)
                if r is not None:
                    n = min(PAGE, data.size - noff)
                    assert L.pbsgpu_memcpy_h2d(eng._h, r[0], data[noff:noff + n].ctypes.data, n) == 0
                    noff += n
                    ring.commit(nsid, n, final=(noff == data.size))
            if ooff < odata.size:
                r = ring.reserve(other)
                if r is not None:
                    n = min(PAGE, odata.size - ooff)
                    assert L.pbsgpu_memcpy_h2d(eng._h, r[0], odata[ooff:ooff + n].ctypes.data, n) == 0
                    ooff += n
                    ring.commit(other, n, final=(ooff == odata.size))
            ring.pump()
            recs, fin = ring.poll(nsid)                      # raises E_DENSITY if the slot inherited the dead stream's failure
            got.append(recs.copy())
            ogot.append(ring.poll(other)[0].copy())
        assert fin, ring.debug()
        _assert_same(np.concatenate(got), O.chunk_and_digest(cfg, data, [(0, data.size)]), ("next stream", k))
        ring.close_stream(nsid)
    t1 = time.time()
    fin = False
    while not fin and time.time() - t1 < 30:
        if ooff < odata.size:
            r = ring.reserve(other)
            if r is not None:
                n = min(PAGE, odata.size - ooff)
                assert L.pbsgpu_memcpy_h2d(eng._h, r[0], odata[ooff:ooff + n].ctypes.data, n) == 0
                ooff += n
                ring.commit(other, n, final=(ooff == odata.size))
        ring.pump()
        recs, fin = ring.poll(other)
        ogot.append(recs.copy())
    assert fin
    _assert_same(np.concatenate(ogot), O.chunk_and_digest(cfg, odata, [(0, odata.size)]), "the bystander")
    ring.close_stream(other)
    ring.quiesce()
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    ring.close()
    eng.close()


def test_quiesce_hashes_rounds_that_were_cut_ahead_of_the_service(gpu_lib, O, monkeypatch):
    """fill -> pump -> quiesce -> poll on an idle ring with ONE bulk stream: the rounds are cut ahead of the service start
    (the lone-stream deferral), so there is no service when quiesce is called. Its contract is "everything enqueued is
    hashed": every chunk the rounds cut must be there afterwards, digest and all."""
    from pbs_plus_amd import PageRing

    monkeypatch.setenv("PBSGPU_RING_LONE_DEFER_MS", "5000")   # the deferral must not time out under a slow test host
    avg = 4096
    eng = _engine(avg)
    PAGE = 65536
    ring = PageRing(eng, arena_bytes=64 * (PAGE + 256), page_bytes=PAGE, max_streams=4, sha_cus=4, round_pages=8)
    total = 40 * PAGE + 321
    first = 24 * PAGE
    sid = ring.open()
    assert ring.fill(sid, 77, 0, first, final=False) == first
    ring.pump()
    st = ring.stats()
    assert st["rounds"] >= 1 and st["service_launches"] == 0, st     # cut ahead: no service yet
    ring.quiesce()
    st = ring.stats()
    assert st["service_launches"] == 1 and st["rounds_done"] == st["rounds"], st
    got, fin = ring.poll(sid)
    assert not fin and got.size > 0
    data = O.fill(total, 77, 0)
    want = O.chunk_and_digest(O.new_config(avg), data, [(0, total)])
    # every chunk that ends inside the bytes fed so far and cannot be the open one is there, digests included
    must = int(np.searchsorted(want["end"], first - 4 * avg * 4, side="right"))
    assert got.size >= must > 10, (got.size, must)
    _assert_same(got, want[:got.size], "after quiesce")
    # ... and the stream goes on from there (a new service launch), to the end
    rest = [got]
    left = total - first
    t0 = time.time()
    while not fin and time.time() - t0 < 30:
        if left:
            left -= ring.fill(sid, 77, 0, left, final=True)
        ring.pump()
        r, fin = ring.poll(sid)
        rest.append(r.copy())
    assert fin
    _assert_same(np.concatenate(rest), want, "whole stream")
    ring.close_stream(sid)
    ring.quiesce()
    ring.close()
    eng.close()


def test_a_stream_longer_than_one_tebibyte(gpu_lib, O):
    """1 TiB + 64 GiB of generated bytes as ONE stream through the default-geometry ring (avg 4 MiB): ~290 k records whose ends
    pass 2^40. Size-independent properties over all of them, and the oracle at restart points spread over the stream —
    including points beyond 2^40 and the tail (the bytes of any range are regenerated from the seed: the ring keeps nothing)."""
    from oracle import restart_check
    from pbs_plus_amd import PageRing

    avg = 4 << 20
    eng = _engine(avg)
    ring = PageRing(eng)
    total = (1 << 40) + 64 * GiB + 12345
    seed = 20260922
    sid = ring.open()
    left, recs, fin = total, [], False
    quota = 256 * ring.page_bytes
    t0 = time.time()
    while not fin:
        if left:
            want = min(left, quota)
            left -= ring.fill(sid, seed, 4, want, final=(want == left))
        ring.pump()
        r, fin = ring.poll(sid, cap=8192)
        if r.size:
            recs.append(r.copy())
        assert time.time() - t0 < 240, ring.debug()
    dt = time.time() - t0
    ring.close_stream(sid)
    ring.quiesce()
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    recs = np.concatenate(recs)
    cfg = O.new_config(avg)
    ends = recs["end"].astype(np.uint64)
    assert int(ends[-1]) == total and np.all(np.diff(ends.astype(np.int64)) > 0)
    assert int(recs["size"].astype(np.uint64).sum()) == total
    assert np.array_equal(np.diff(np.concatenate([[0], ends.astype(np.int64)])), recs["size"].astype(np.int64))
    assert np.all(recs["size"][:-1] >= cfg.min) and np.all(recs["size"] <= cfg.max)
    assert int((ends > (1 << 40)).sum()) > 10_000
    recs = recs.copy()
    recs["segment"] = 0
    res = restart_check.check_batch(lambda off, n: O.fill(n, seed, 4, stream_off=off), None, recs, avg, nbytes=total, k=34,
                                    span=64 << 20)
    assert res["ok"], res
    assert res["max_offset"] == total and res["records_checked"] > 400, res
    print(f"1.06 TiB stream: {recs.size} records in {dt:.1f} s ({total / dt / GiB:.0f} GiB/s), {res['records_checked']} records "
          f"re-checked at {res['points']} restart points up to offset {res['max_offset']}")
    ring.close()
    eng.close()


def test_parked_frees_are_bounded_and_the_ring_survives_the_service_restarts(gpu_lib, O, monkeypatch):
    """While a ring's service runs, hipFree would wait for it: frees are parked. Round 4 parked them without bound (a caller
    that frees and reallocates under a busy ring ran out of memory). Now the parked bytes have a cap; beyond it the ring
    parks its service, the memory is really freed, the next round starts the service again. 48 x 8 GiB allocated and freed
    under a ring that is hashing a stream all the while: more than the device holds unless the frees really happen."""
    import gc

    from pbs_plus_amd import PageRing

    gc.collect()
    monkeypatch.setenv("PBSGPU_GRAVEYARD_MIB", "20000")
    avg = 65536
    eng = _engine(avg)
    PAGE = 262144
    ring = PageRing(eng, arena_bytes=256 * (PAGE + 256), page_bytes=PAGE, max_streams=4, sha_cus=8, round_pages=16)
    total = 96 * PAGE * 8 + 4321
    sid = ring.open()
    left, got, fin = total, [], False
    step = total // 48 // PAGE * PAGE
    for i in range(48):
        if left:
            want = min(left, step)
            left -= ring.fill(sid, 808, 0, want, final=False)
        ring.pump()
        buf = eng.alloc(8 * GiB)                             # E_NOMEM here = parked frees were never really freed
        buf.free()
        r, fin = ring.poll(sid)
        got.append(r.copy())
    t0 = time.time()
    while not fin and time.time() - t0 < 60:
        if left:
            left -= ring.fill(sid, 808, 0, left, final=True)
        ring.pump()
        r, fin = ring.poll(sid)
        got.append(r.copy())
    assert fin, ring.debug()
    st = ring.stats()
    assert st["service_launches"] >= 2, st                    # the service was parked for the frees at least once
    _assert_same(np.concatenate(got), O.chunk_and_digest(O.new_config(avg), O.fill(total, 808, 0), [(0, total)]), "stream")
    ring.close_stream(sid)
    ring.close()
    eng.close()


def test_digest_set_reduce_reports_bad_arguments_collectively(gpu_lib, O):
    """n > cap_records used to return from the calling rank alone, BEFORE the all-gather the other ranks were already waiting
    in. Now the ranks agree first (64-byte all-gathers): the call fails on every rank with the same error, and the
    communicator is still good for the next, well-formed call. (One rank here; the agreement path is the same code.)"""
    import gc

    from pbs_plus_amd import Comm, PbsGpuError, RECORD_DTYPE, _lib

    gc.collect()
    eng = _engine(4096)
    rng = np.random.default_rng(3)
    n = 5000
    recs = np.zeros(n, dtype=RECORD_DTYPE)
    recs["digest"] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    recs["digest"][n // 2:] = recs["digest"][: n - n // 2]
    recs["size"] = 4096
    recs["end"] = np.cumsum(recs["size"].astype(np.uint64))
    comm = Comm(eng, Comm.unique_id(), 0, 1)
    with pytest.raises(PbsGpuError) as ei:
        comm.dedup(recs, n - 1)                              # more records than the capacity the ranks agreed on
    assert ei.value.status == _lib.E_INVALID
    with pytest.raises(PbsGpuError) as ei:
        comm.dedup(recs, 0)
    assert ei.value.status == _lib.E_INVALID
    want_dup, want_stats = eng.dedup(recs)
    dup, stats = comm.dedup(recs, 4 * n)                      # only max(n) records travel, whatever the capacity
    assert np.array_equal(dup, want_dup) and stats == want_stats
    comm.close()
    eng.close()


def test_service_split_follows_the_share_of_bytes_in_long_chunks(gpu_lib, O):
    """The ring starts with 16 express CUs (right for random data: 5 % of its bytes sit in chunks >= 13/16 of the maximum).
    A corpus of zero runs is ALL max-size chunks: every page is then held for a full chain, and the express form's chain is
    1.37x shorter. Every service start re-balances the two services from what the ring has published since the last
    decision; the records stay bit-exact whatever the split (zero runs through express pairs, random files through both)."""
    from pbs_plus_amd import PageRing

    avg = 4 << 20
    eng = _engine(avg)
    ring = PageRing(eng)
    cfg = O.new_config(avg)
    xp0, long_from = ring.express()
    pair0 = ring.stats()["sha_cus"]
    assert xp0 == 16 and long_from == (16 << 20) * 13 // 16, (xp0, long_from)
    fsz = 256 << 20
    zero_jobs = [(1000 + i, 1, fsz + 4096 * i) for i in range(64)]          # 16 GiB of zero runs: nothing but 16 MiB chunks
    got = ring.ingest_synthetic(zero_jobs, timeout_s=120.0, concurrent=32)
    ring.quiesce()
    assert ring.express()[0] == xp0                                        # (a running service is never re-balanced)
    want0 = O.chunk_and_digest(cfg, O.fill(zero_jobs[5][2], zero_jobs[5][0], 1), [(0, zero_jobs[5][2])])
    _assert_same(got[5], want0, "zero file, default split")
    # the next service start sees 100 % of the bytes in long chunks
    mixed = [(2000, 4, fsz + 77), (2001, 1, fsz), (2002, 3, fsz + 12345), (2003, 1, 3 * fsz)]
    got = ring.ingest_synthetic(mixed, timeout_s=120.0)
    xp1 = ring.express()[0]
    st = ring.stats()
    assert xp1 >= 96 and xp1 % 8 == 0 and st["sha_cus"] + xp1 == pair0 + xp0, (xp1, st)
    for j, g in zip(mixed, got):
        _assert_same(g, O.chunk_and_digest(cfg, O.fill(j[2], j[0], j[1]), [(0, j[2])]), ("after re-balancing", j))
    ring.quiesce()
    # ... and back: 48 GiB of random-like files dominate the (halved) memory of the zero runs
    rnd = [(3000 + i, 4, 2 * fsz) for i in range(96)]
    got = ring.ingest_synthetic(rnd, timeout_s=120.0, concurrent=32)
    ring.quiesce()
    _assert_same(got[7], O.chunk_and_digest(cfg, O.fill(rnd[7][2], rnd[7][0], 4), [(0, rnd[7][2])]), "random file")
    got = ring.ingest_synthetic([(4000, 4, fsz)], timeout_s=60.0)
    xp2 = ring.express()[0]
    assert 16 <= xp2 < xp1 and ring.stats()["sha_cus"] + xp2 == pair0 + xp0, (xp1, xp2)
    _assert_same(got[0], O.chunk_and_digest(cfg, O.fill(fsz, 4000, 4), [(0, fsz)]), "after re-balancing back")
    ring.quiesce()
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    ring.close()
    eng.close()


@pytest.mark.parametrize("staged", [True, False])
def test_round_tables_staged_or_read_from_host_memory(gpu_lib, O, monkeypatch, staged):
    """The round's first kernel (k_ring_stage) copies the host-written page / segment / record-base tables into device memory and
    every other kernel of the round reads the copy; PBSGPU_RING_STAGE_INPUTS=0 keeps rounds 3-4's reads of the mapped host memory.
    Same streams, same records either way: nine streams of every generator kind (empty, 1-byte, 63-byte, page-exact ones among
    them) through a 24-page ring, two at a time, so that pages, slots and input blocks are all re-used many times."""
    from pbs_plus_amd import PageRing

    if not staged:
        monkeypatch.setenv("PBSGPU_RING_STAGE_INPUTS", "0")
    avg = 4096
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=24 * (65536 + 256), page_bytes=65536, max_streams=4, sha_cus=4, round_pages=6)
    jobs = [(921, 0, (1 << 20) + 5), (922, 1, 300 * 1024), (923, 3, 700 * 1024 + 3), (924, 0, 0), (925, 0, 63), (926, 2, 65536),
            (927, 0, 65536 * 3), (928, 4, 65536 * 2 + 1), (929, 0, 1)]
    got = ring.ingest_synthetic(jobs, timeout_s=60.0, concurrent=2)
    ring.quiesce()
    cfg = O.new_config(avg)
    for j, g in zip(jobs, got):
        want = O.chunk_and_digest(cfg, O.fill(j[2], j[0], j[1]), [(0, j[2])]) if j[2] else np.zeros(0, dtype=O.RECORD_DTYPE)
        _assert_same(g, want, ("staged" if staged else "host tables", j))
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"] and st["pages_recycled"] == st["pages_enqueued"], st
    ring.close()
    eng.close()


@pytest.mark.parametrize("light_load_rule", [True, False])
def test_one_file_alone_sends_its_longer_chunks_express(gpu_lib, O, monkeypatch, light_load_rule):
    """Under load the express service takes chunks of at least 13/16 of the maximum (what its 16 CUs can keep up with); while
    fewer than 3/4 of its lane pairs are taken — one file alone on an idle ring — chunks from 11/16 of the maximum go express
    too, so that the file's last record waits for the express chain of its longest chunk and not for the pair chain of one
    just below the threshold. Which queue a chunk took is visible in the long queue's tail (cumulative); the records are the
    oracle's either way."""
    import re
    from pbs_plus_amd import PageRing

    if not light_load_rule:
        monkeypatch.setenv("PBSGPU_RING_LONG_LO_BYTES", "0")
    avg = 4 << 20
    eng = _engine(avg)
    ring = PageRing(eng)
    assert ring.express() == (16, (16 << 20) * 13 // 16)
    n = 3 * GiB + 4096            # 827 chunks: 13 of at least 13/16 of the maximum, 19 of at least 11/16
    got = ring.ingest_synthetic([(515, 4, n)], timeout_s=120.0)[0]
    ring.quiesce()
    ltail = int(re.search(r"ltail=(\d+)", ring.debug()).group(1))
    n13 = int((got["size"] >= (16 << 20) * 13 // 16).sum())
    n11 = int((got["size"] >= (16 << 20) * 11 // 16).sum())
    assert n11 > n13 > 0, (n11, n13)                      # (the file has chunks between the two thresholds)
    assert ltail == (n11 if light_load_rule else n13), (ltail, n11, n13)
    _assert_same(got, O.chunk_and_digest(O.new_config(avg), O.fill(n, 515, 4), [(0, n)]), "one file alone")
    ring.close()
    eng.close()
