"""Round-4 GPU tests (through the C ABI): the page ring as the ONE streaming engine — suggested boundaries and the
reader-buffer rule in ring rounds, payload streams as clients of the engine's ring (sections, slot reuse, a failing stream
beside healthy ones, device-wide synchronisation right after finish) — and the multi-GPU digest-set reduce behind the C ABI
(pbsgpu_comm_*: RCCL resolved by libpbsgpu itself), with one rank on the one GPU of the test box."""
import json
import os
import subprocess
import sys
import threading
import time

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _engine(avg, inflight=1):
    from pbs_plus_amd import Engine, buzhash

    return Engine(buzhash.NewConfig(avg), device=0, inflight=inflight)


def _feed_ring_host_bytes(eng, ring, sid, data, page, suggestions=()):
    """reserve / H2D / commit the whole stream page by page, announcing every suggested boundary ahead of its bytes"""
    L = eng._L
    sugg = list(suggestions)
    got, off, fin = [], 0, False
    t0 = time.time()
    while not fin and time.time() - t0 < 60:
        if off < data.size:
            r = ring.reserve(sid)
            if r is not None:
                n = min(page, data.size - off)
                while sugg and sugg[0] <= off + n + (1 << 30):      # announced ahead of the data (all of them: they are known)
                    ring.suggest(sid, sugg.pop(0))
                assert L.pbsgpu_memcpy_h2d(eng._h, r[0], data[off:off + n].ctypes.data, n) == 0
                off += n
                ring.commit(sid, n, final=(off == data.size))
        ring.pump()
        recs, fin = ring.poll(sid)
        got.append(recs.copy())
    assert fin, ring.debug()
    return np.concatenate(got)


@pytest.mark.parametrize("feed,absolute", [(1, False), (4096, False), (65536, True), (0, False), (1000, True)])
def test_ring_rounds_honour_suggested_boundaries_and_the_reader_buffer_rule(gpu_lib, O, feed, absolute):
    """pbsgpu_ring_suggest: a stream cut in MANY rounds (64 KiB pages, the open chunk carried in device state, only new
    pages scanned) gives the payload chunker's cut list — byte-serial rule and reader-buffer rule on both grids, where a
    boundary beyond the bytes seen so far pre-empts a hash cut that lies in pages no later round scans again."""
    from pbs_plus_amd import PageRing

    cfg = O.new_config(4096)
    rng = np.random.default_rng(4321)
    n = 2_000_003
    data = O.fill(n, 177, 0)
    data[300_000:480_000] = 0
    plain = O.chunk_stream(cfg, data)
    near = [int(e) + int(d) for e, d in zip(plain[5:300:5], rng.integers(1, 3000, 300))]
    sugg = sorted(set(x for x in near + [int(x) for x in rng.integers(1, n, 120)] + [n, 64, 1024, 1025, 65536, 131072] if x <= n))
    want_ends = O.chunk_stream_suggested(cfg, data, sugg, feed=feed, absolute=absolute)
    if feed != 1:
        assert not np.array_equal(want_ends, O.chunk_stream_suggested(cfg, data, sugg, feed=1)), "the input must make the feed matter"
    eng = _engine(4096)
    eng.set_suggested_feed(feed, absolute)
    ring = PageRing(eng, arena_bytes=48 * (65536 + 256), page_bytes=65536, max_streams=4, sha_cus=4, round_pages=4)
    sid = ring.open()
    got = _feed_ring_host_bytes(eng, ring, sid, data, 65536, sugg)
    ring.close_stream(sid)
    ring.quiesce()
    assert np.array_equal(got["end"], want_ends), (feed, absolute, got["end"][:8], want_ends[:8])
    starts = np.concatenate([[0], want_ends[:-1]]).astype(np.int64)
    for i in (0, len(want_ends) // 2, len(want_ends) - 1):
        assert bytes(got["digest"][i]) == O.sha256(data[int(starts[i]):int(want_ends[i])])
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    ring.close()
    eng.close()


def test_payload_streams_share_the_engine_ring_sections_and_slot_reuse(gpu_lib, O, monkeypatch):
    """pbsgpu_stream_* on the engine's ring with only FOUR ring-stream slots: one archive with dozens of InjectChunks
    sections (each a ring stream of its own, most of them shorter than a page), a second archive written from another
    thread at the same time — sections wait for slots, records come out in stream order with the right section numbers and
    payload positions, every byte accounted for."""
    from pbs_plus_amd import PayloadStream

    monkeypatch.setenv("PBSGPU_STREAM_RING_SLOTS", "4")
    monkeypatch.setenv("PBSGPU_STREAM_RING_GIB", "0.02")
    eng = _engine(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(7)
    secs = [O.fill(int(rng.integers(0, 200_000)), 900 + i, i % 4) for i in range(40)]
    injs = [int(rng.integers(0, 50_000)) for _ in secs]
    other = O.fill(3_000_000, 5555, 0)
    res = {}

    def second():
        rng2 = np.random.default_rng(8)
        ps = PayloadStream(eng)
        pos = 0
        while pos < other.size:
            m = min(int(rng2.integers(1, 300_000)), other.size - pos)
            ps.write(other[pos:pos + m])
            pos += m
        ps.finish()
        res["other"] = ps.poll()
        ps.close()

    th = threading.Thread(target=second)
    th.start()
    ps = PayloadStream(eng)
    got = []
    for d, inj in zip(secs, injs):
        ps.write(d)
        ps.inject(inj)
        got.append(ps.poll())
    ps.finish()
    got.append(ps.poll())
    th.join(timeout=120)
    assert not th.is_alive()
    got = np.concatenate(got)
    want_end, want_dig, want_seg, base = [], [], [], 0
    for k, (d, inj) in enumerate(zip(secs, injs)):
        if d.size:
            w = O.chunk_and_digest(cfg, d, [(0, d.size)])
            want_end.append(w["end"] + np.uint64(base))
            want_dig.append(w["digest"])
            want_seg += [k] * w.size
        base += d.size + inj
    assert np.array_equal(got["end"], np.concatenate(want_end))
    assert np.array_equal(got["digest"], np.concatenate(want_dig))
    assert got["segment"].tolist() == want_seg
    assert ps.position() == base
    wo = O.chunk_and_digest(cfg, other, [(0, other.size)])
    assert np.array_equal(res["other"]["end"], wo["end"]) and np.array_equal(res["other"]["digest"], wo["digest"])
    ps.close()
    eng.close()


def test_a_dense_payload_stream_is_exact_and_the_device_is_free_after_finish(gpu_lib, O):
    """One archive carries a crafted 64-byte period (more candidates than a scan tile has slots): rounds 4-5 answered ITS
    calls with PBSGPU_E_DENSITY; since round 6 it is cut exactly (on-demand re-scan, DenseTiles) like the ordinary archive
    written through the same engine at the same time. And after finish() of the last stream the ring's persistent service
    has been parked: a device-wide synchronisation (what a host does before freeing memory or handing the GPU to someone
    else) returns at once instead of waiting for a kernel that only ends on request."""
    import ctypes as C

    from pbs_plus_amd import PayloadStream

    eng = _engine(4096)
    hip = C.CDLL("libamdhip64.so.7", mode=os.RTLD_NOLOAD)         # the runtime instance libpbsgpu.so is linked against
    hip.hipDeviceSynchronize.restype = C.c_int
    cfg = O.new_config(4096)
    pat = None
    rng = np.random.default_rng(5)
    for _ in range(20000):
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(p, 8)).size >= 6:
            pat = p
            break
    if pat is None:
        pytest.skip("no dense pattern found")
    bad = np.concatenate([O.fill(100_000, 9, 0), np.tile(pat, 4 * 65536 // 64)])
    good = O.fill(2_500_000, 10, 3)
    ok = PayloadStream(eng)
    ko = PayloadStream(eng)
    pos = 0
    while pos < bad.size:
        ko.write(bad[pos:pos + 50_000])
        ok.write(good[pos:pos + 50_000])
        pos += 50_000
    ko.finish()
    got_bad = ko.poll()
    want_bad = O.chunk_and_digest(cfg, bad, [(0, bad.size)])
    assert np.array_equal(got_bad["end"], want_bad["end"]) and np.array_equal(got_bad["digest"], want_bad["digest"])
    ok.write(good[min(pos, good.size):])
    ok.finish()
    got = ok.poll()
    want = O.chunk_and_digest(cfg, good, [(0, good.size)])
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    t0 = time.time()
    assert hip.hipDeviceSynchronize() == 0
    assert time.time() - t0 < 1.0, "a device-wide synchronisation after finish() waited for the ring's service"
    ko.close()
    ok.close()
    ps = PayloadStream(eng)
    ps.write(good[:700_000])
    ps.finish()
    w2 = O.chunk_and_digest(cfg, good[:700_000], [(0, 700_000)])
    g2 = ps.poll()
    assert np.array_equal(g2["end"], w2["end"]) and np.array_equal(g2["digest"], w2["digest"])
    ps.close()
    eng.close()


def test_comm_digest_set_reduce_behind_the_c_abi_one_rank(gpu_lib, O):
    """pbsgpu_comm_unique_id / pbsgpu_comm_create / pbsgpu_digest_allgather_dedup with world size 1 on the one GPU here:
    libpbsgpu dlopens RCCL, builds the communicator, runs ncclAllGather of the [count | records] slot and the device dedup;
    flags and statistics equal pbsgpu_dedup_host on the same records. (N > 1 over xGMI is the driver's to run: bench.py
    repeats this reduce over all ranks beside the torch.distributed path and reports whether they agree.)"""
    import gc

    from pbs_plus_amd import Comm, RECORD_DTYPE

    gc.collect()   # engines of earlier tests that are only waiting for the collector still hold their page rings (48 GiB each)
    eng = _engine(4096)
    rng = np.random.default_rng(11)
    n = 30_000
    recs = np.zeros(n, dtype=RECORD_DTYPE)
    recs["digest"] = rng.integers(0, 256, (n, 32), dtype=np.uint8)
    recs["size"] = rng.integers(1, 1 << 20, n)
    recs["end"] = np.cumsum(recs["size"].astype(np.uint64))
    dupsrc = rng.integers(0, n // 2, n // 3)
    recs["digest"][n - n // 3:] = recs["digest"][dupsrc]           # planted duplicates
    want_dup, want_stats = eng.dedup(recs)
    comm = Comm(eng, Comm.unique_id(), 0, 1)
    for cap in (n, n + 1234):
        dup, stats = comm.dedup(recs, cap)
        assert np.array_equal(dup, want_dup) and stats == want_stats, (stats, want_stats)
    dup0, stats0 = comm.dedup(recs[:0], 16)
    assert dup0.size == 0 and stats0["nrecords"] == 0
    # records that already are in DEVICE memory (round 6): no host round trip, same answer
    dbuf = eng.alloc(recs.nbytes)
    dbuf.upload(recs.view(np.uint8))
    dupd, statsd = comm.dedup_device(dbuf.ptr, n, n + 5)
    assert np.array_equal(dupd, want_dup) and statsd == want_stats, (statsd, want_stats)
    dbuf.free()
    comm.close()
    eng.close()


def test_single_stream_split_behind_the_c_abi_one_rank(gpu_lib, O):
    """pbsgpu_comm_split_stream (round 6; rounds 2-5: Python over torch.distributed): one stream cut and hashed through the
    communicator's two exchange steps — candidates of the own range, digests of the chunks that start in it — with world size
    1 on the one GPU here (scan -> all-gather -> resolve -> hash -> all-gather: the same code as with N ranks; the two-rank
    arithmetic is checked on the CPU, tests/test_host_logic.py). Bit-exact vs the oracle, incl. a dense stretch and a tail
    shorter than a block."""
    import gc

    from pbs_plus_amd import Comm
    from pbs_plus_amd.engine import split_plan

    gc.collect()
    for avg, n in ((4096, 3_000_017), (4 << 20, (96 << 20) + 5)):
        eng = _engine(avg)
        cfg = O.new_config(avg)
        data = O.fill(n, 77, 3)
        assert split_plan(n, 1, 0, cfg.max) == (0, n, 0, n)
        buf = eng.alloc(n + 64)
        buf.upload(data)
        comm = Comm(eng, Comm.unique_id(), 0, 1)
        got = comm.split_stream(buf.ptr, n)
        want = O.chunk_and_digest(cfg, data, [(0, n)])
        assert got.size == want.size and np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
        assert comm.split_stream(0, 0).size == 0
        comm.close()
        buf.free()
        eng.close()


def test_bench_ring_forced_dist_runs_the_c_abi_reduce_beside_torch(gpu_lib):
    """bench.py's multi-rank branch with ONE rank over RCCL (PBS_BENCH_FORCE_DIST): the digest-set reduce of the timed region
    goes through libpbsgpu's own communicator, and equals the torch.distributed path's."""
    env = dict(os.environ, PBS_BENCH_FORCE_DIST="1", MASTER_PORT="29577")
    for k in ("RANK", "LOCAL_RANK", "WORLD_SIZE"):
        env.pop(k, None)
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gib", "0.25", "--steps", "3", "--warmup", "1",
                          "--avg", str(1 << 20), "--arena-gib", "1.5", "--ring-streams", "2", "--no-cpu-baseline"],
                         env=env, capture_output=True, text=True, timeout=300)
    assert out.returncode == 0, out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    c = d["results"]["c_abi_digest_reduce"]
    assert c["ok"] is True and c["equals_torch_path"] is True and c["stats"]["nrecords"] > 0, c
    # round 6: the reduce INSIDE the timed region is the C ABI's (libpbsgpu's own RCCL communicator), torch is the cross-check
    assert c["timed_through_c_abi"] is True and "pbsgpu_digest_allgather_dedup" in c["timed_path"], c


def test_ring_piece_table_producer_matches_the_host_rebuild(gpu_lib, O):
    """pbsgpu_ring_fill_pieces (the synthetic producer of BASELINE configs[4] through the ring): a stream defined by a piece
    table over generator 4 — kept extents of a base file interleaved with new bytes, pieces crossing page edges, a short
    last page — is cut and hashed exactly like the same bytes rebuilt on the host from the oracle's generator twin."""
    from pbs_plus_amd import PageRing

    eng = _engine(4096)
    cfg = O.new_config(4096)
    rng = np.random.default_rng(3)
    rows, pos, npos, src = [], 0, 0, 0
    for i in range(60):
        ln = int(rng.integers(1, 6000)) * 16
        if i % 3 == 1:                                   # new bytes
            rows.append((pos, ln, npos, 777))
            npos += ln
        else:                                            # kept extent (an occasional deletion in between)
            src += int(rng.integers(0, 300)) * 16 * (i % 2)
            rows.append((pos, ln, src, 555))
            src += ln
        pos += ln
    host = np.empty(pos, dtype=np.uint8)
    for dst, ln, so, seed in rows:
        O.fill(ln, seed, 4, stream_off=so, out=host[dst:dst + ln])
    want = O.chunk_and_digest(cfg, host, [(0, pos)])
    ring = PageRing(eng, arena_bytes=64 * (65536 + 256), page_bytes=65536, max_streams=4, sha_cus=4, round_pages=6)
    sid = ring.open()
    left, first, got, fin = pos, True, [], False
    t0 = time.time()
    while not fin and time.time() - t0 < 60:
        if left:
            want_n = min(left, 5 * 65536)
            left -= ring.fill_pieces(sid, np.array(rows, dtype=np.uint64) if first else None, want_n, final=(want_n == left))
            first = False
        ring.pump()
        recs, fin = ring.poll(sid)
        got.append(recs.copy())
    assert fin
    got = np.concatenate(got)
    assert np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
    ring.close_stream(sid)
    ring.quiesce()
    ring.close()
    eng.close()
