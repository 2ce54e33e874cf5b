"""The page ring (pbsgpu_ring_*: page-granular memory release + persistent cross-stream SHA-256 service) against the
CPU oracle, through the C ABI. Every stream's records (end, size, digest) must be bit-identical to the serial
chunker + SHA-256 over the same bytes — whatever pages the stream's bytes landed in, however its chunks were spread
over rounds, and however often the service kernel was stopped and started. Mirrors the shapes the reference's writers
produce: many archives at once (internal/pxarmount/commit_reuse.go:457, internal/tapeio/converter.go:836), empty and
tiny files (commit_walk_test.go:21-147), zero runs."""
import ctypes as C
import json
import os
import subprocess
import sys
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

GiB = 1 << 30
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(autouse=True)
def _short_idle_timeout(monkeypatch):
    # a service wave that sees no work gives up after this long: a bug must fail a test, not hang the box
    monkeypatch.setenv("PBSGPU_RING_IDLE_TIMEOUT_S", "5")


def _engine(avg):
    from pbs_plus_amd import Engine, buzhash

    return Engine(buzhash.NewConfig(avg), device=0, inflight=1)


def _oracle_records(O, avg, jobs):
    cfg = O.new_config(avg)
    out = [None] * len(jobs)

    def one(i):
        seed, kind, n = jobs[i]
        out[i] = O.chunk_and_digest(cfg, O.fill(n, seed, kind), [(0, n)]) if n else np.zeros(0, dtype=O.RECORD_DTYPE)

    ths = [threading.Thread(target=one, args=(i,)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return out


def _assert_same(got, want, what):
    assert got.size == want.size, (what, got.size, want.size)
    assert np.array_equal(got["end"], want["end"]), what
    assert np.array_equal(got["size"], want["size"]), what
    assert np.array_equal(got["digest"], want["digest"]), what


SMALL = dict(page_bytes=65536, max_streams=8, sha_cus=4, round_pages=6)


@pytest.mark.parametrize("name,avg,opt,jobs,conc", [
    ("one stream", 4096, dict(arena_bytes=24 * (65536 + 256), **SMALL), [(11, 0, 200 * 1024 + 17)], None),
    ("mixed, empty, tiny, page-exact", 4096, dict(arena_bytes=24 * (65536 + 256), **SMALL),
     [(21, 0, (1 << 20) + 5), (22, 1, 300 * 1024), (23, 3, 700 * 1024 + 3), (24, 0, 0), (25, 0, 63), (26, 2, 65536),
      (27, 0, 65536 * 3), (28, 4, 65536 * 2 + 1), (29, 0, 1)], None),
    ("two at a time through 10 pages", 4096, dict(arena_bytes=10 * (65536 + 256), page_bytes=65536, max_streams=2, sha_cus=2,
                                                  round_pages=3),
     [(31, 0, (1 << 20) + 5), (32, 1, 300 * 1024), (33, 3, 700 * 1024 + 3), (34, 0, 64), (35, 0, 65), (36, 4, 131072)], 2),
    ("avg 64 KiB", 65536, dict(arena_bytes=96 * (262144 + 256), page_bytes=262144, max_streams=8, sha_cus=16, round_pages=16),
     [(41 + i, i % 5, (8 << 20) + 4099 * i) for i in range(6)], None),
    # PBSGPU_RING_F_DENSE_SERVICE = 64: the pair service with eight waves per CU (measured and not the default: DESIGN.md 5.2)
    # the LANES service (lanes_cus of the pair service's CUs, chunks of at most short_bytes: DESIGN.md 5.5): 2 of 4 CUs, and every
    # chunk short enough for it
    ("lanes service", 4096, dict(arena_bytes=24 * (65536 + 256), lanes_cus=2, short_bytes=8192, **SMALL),
     [(61, 0, (1 << 20) + 5), (62, 1, 300 * 1024), (63, 3, 700 * 1024 + 3), (64, 0, 0), (65, 0, 63), (66, 2, 65536)], None),
    ("lanes service, all chunks", 4096, dict(arena_bytes=24 * (65536 + 256), lanes_cus=3, short_bytes=16384, **SMALL),
     [(71, 0, (1 << 20) + 5), (72, 4, 300 * 1024), (73, 3, 700 * 1024 + 3), (74, 0, 1)], None),
    # PBSGPU_RING_F_DENSE_LANES = 128: the lanes service with eight waves per CU (profiles/r06_lanes_service.log, calls 37-40)
    ("lanes service, eight waves per CU", 4096, dict(arena_bytes=24 * (65536 + 256), lanes_cus=2, short_bytes=16384, flags=128, **SMALL),
     [(81, 0, (1 << 20) + 5), (82, 1, 300 * 1024), (83, 3, 700 * 1024 + 3), (84, 0, 0), (85, 0, 65), (86, 2, 65536)], None),
    ("dense service form", 4096, dict(arena_bytes=24 * (65536 + 256), flags=64, **SMALL),
     [(51, 0, (1 << 20) + 5), (52, 1, 300 * 1024), (53, 3, 700 * 1024 + 3), (54, 0, 0), (55, 0, 63), (56, 2, 65536)], None),
])
def test_ring_streams_match_the_oracle(gpu_lib, O, name, avg, opt, jobs, conc):
    from pbs_plus_amd import PageRing

    eng = _engine(avg)
    ring = PageRing(eng, **opt)
    got = ring.ingest_synthetic(jobs, timeout_s=60.0, concurrent=conc)
    ring.quiesce()
    st = ring.stats()
    want = _oracle_records(O, avg, jobs)
    for i, (g, w) in enumerate(zip(got, want)):
        _assert_same(g, w, (name, i, jobs[i]))
    # every page came back, exactly once per use
    assert st["pages_free"] == st["pages_total"] and st["pages_recycled"] == st["pages_enqueued"], st
    assert st["service_launches"] == 1
    ring.close()
    eng.close()


def test_ring_production_chunker_three_streams(gpu_lib, O):
    """avg 4 MiB (buzhash.NewConfig(4 << 20), commit_orchestrate.go:144): default page size (61 scan tiles, 16.2 MiB),
    random / 30 % zero extents / all-zero streams of 0.5-1.5 GiB through 3 GiB of pages — the arena turns over"""
    from pbs_plus_amd import PageRing

    avg = 4 << 20
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=3 * GiB, max_streams=4, sha_cus=64, round_pages=64)
    jobs = [(51, 0, 3 * GiB // 2 + 56), (52, 3, 3 * GiB // 2), (53, 1, GiB // 2 + 4096), (54, 4, GiB + 24)]
    got = ring.ingest_synthetic(jobs, timeout_s=120.0)
    ring.quiesce()
    st = ring.stats()
    want = _oracle_records(O, avg, jobs)
    for i, (g, w) in enumerate(zip(got, want)):
        _assert_same(g, w, (i, jobs[i]))
    assert st["page_bytes"] == 61 * 64 * 34 * 128 and st["pages_free"] == st["pages_total"]
    assert st["pages_enqueued"] > st["pages_total"]          # the arena turned over
    assert (got[2]["size"][:-1] == 16 << 20).all()           # a zero run is cut at max size only
    ring.close()
    eng.close()


def test_ring_reserve_commit_external_producer_and_service_restart(gpu_lib, O):
    """The caller's own producer writes the pages (here: pbsgpu_fill_device / an H2D copy into the reserved pointer),
    commit hands them over; between two groups of streams the ring is quiesced (service kernel stopped: the device is
    idle, hipDeviceSynchronize returns) and started again by the next pump."""
    from pbs_plus_amd import PageRing

    avg = 65536
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=40 * (262144 + 256), page_bytes=262144, max_streams=4, sha_cus=8, round_pages=8)
    page = ring.page_bytes
    L = eng._L

    def feed(jobs):
        res = {}
        streams = {ring.open(): [j, 0] for j in range(len(jobs))}
        import time
        t0 = time.time()
        while streams:
            for sid in list(streams):
                j, off = streams[sid]
                seed, kind, n, host = jobs[j]
                if off < n or (n == 0 and off == 0):
                    if n == 0:
                        ring.commit(sid, 0, final=True)
                        streams[sid][1] = 1
                        continue
                    r = ring.reserve(sid)
                    if r is not None:
                        ptr, cap = r
                        m = min(cap, n - off)
                        if host is None:
                            eng.fill(ptr, (m + 7) & ~7, seed, kind, stream_off=off)   # the page has room for the rounding
                        else:
                            assert L.pbsgpu_memcpy_h2d(eng._h, ptr, host[off:off + m].ctypes.data, m) == 0
                        ring.commit(sid, m, final=(off + m == n))
                        streams[sid][1] = off + m
            ring.pump()
            for sid in list(streams):
                recs, fin = ring.poll(sid)
                if recs.size:
                    res.setdefault(streams[sid][0], []).append(recs.copy())
                if fin:
                    ring.close_stream(sid)
                    del streams[sid]
            assert time.time() - t0 < 60, ring.stats()
        return [np.concatenate(res[j]) if j in res else np.zeros(0, dtype=O.RECORD_DTYPE) for j in range(len(jobs))]

    rng = np.random.default_rng(3)
    host_bytes = rng.integers(0, 256, 3 * page + 77, dtype=np.uint8)
    host_bytes[page - 40:page + 40] = 0                     # a constant run across a page boundary
    groups = [[(61, 0, 5 * page + 1234, None), (62, 3, 2 * page, None), (0, 0, host_bytes.size, host_bytes), (63, 0, 0, None)],
              [(64, 4, 7 * page + 9, None), (65, 1, page + 1, None)]]
    for gi, jobs in enumerate(groups):
        got = feed(jobs)
        ring.quiesce()
        # the persistent kernel has really ended: hipFree waits for the WHOLE device, so freeing a scratch buffer only
        # returns when nothing is running any more (a service that kept running would block it forever)
        done = threading.Event()

        def idle_probe():
            eng.alloc(4096).free()
            done.set()

        threading.Thread(target=idle_probe, daemon=True).start()
        assert done.wait(20.0), "device not idle after quiesce: the service kernel is still running"
        for j, (seed, kind, n, host) in enumerate(jobs):
            data = host if host is not None else O.fill(n, seed, kind)
            want = O.chunk_and_digest(O.new_config(avg), data, [(0, n)]) if n else np.zeros(0, dtype=O.RECORD_DTYPE)
            _assert_same(got[j], want, (gi, j))
        assert ring.stats()["service_launches"] == gi + 1
    ring.close()
    eng.close()


def test_ring_dense_candidates_are_cut_exactly_beside_ordinary_streams(gpu_lib, O):
    """64-byte periodic bytes whose window hash passes the break test put a candidate into every period: 512 per 32 KiB scan
    tile, twice the ring's slots. Rounds 3-5 failed the stream that owned such a tile (PBSGPU_E_DENSITY); since round 6 the
    control kernel re-scans the tile on demand for the one candidate the cut rule needs (DenseTiles, kernels.h) and the
    stream comes out bit-exact like the three ordinary streams that share the ring and the rounds with it. More crafted
    shapes: tests/test_gpu_dense.py."""
    import time

    from pbs_plus_amd import PageRing

    avg = 4096
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=40 * (65536 + 256), **dict(SMALL, max_streams=6))
    cfg = O.new_config(avg)
    pat = None
    rng = np.random.default_rng(5)
    for _ in range(20000):                                   # a 64-byte pattern whose (periodic) window hash passes the break test
        p = rng.integers(0, 256, 64, dtype=np.uint8)
        if O.candidates(cfg, np.tile(p, 8)).size >= 6:     # one candidate per 64-byte period = 43x the nominal density
            pat = p
            break
    if pat is None:
        pytest.skip("no dense pattern found")
    good = [O.fill(5 * 65536 + 1234 * i, 70 + i, i % 4) for i in range(3)]
    bad = np.concatenate([O.fill(65536, 9, 0), np.tile(pat, 3 * 65536 // 64)])   # an ordinary first page, then the crafted period
    datas = good + [bad]
    sids = [ring.open() for _ in datas]
    offs = [0] * len(datas)
    got = [[] for _ in datas]
    done = [False] * len(datas)
    L = eng._L
    t0 = time.time()
    while not all(done) and time.time() - t0 < 60:
        for i, (sid, d) in enumerate(zip(sids, datas)):
            if done[i]:
                continue
            if offs[i] < d.size:
                r = ring.reserve(sid)
                if r is not None:
                    n = min(65536, d.size - offs[i])
                    assert L.pbsgpu_memcpy_h2d(eng._h, r[0], d[offs[i]:offs[i] + n].ctypes.data, n) == 0
                    offs[i] += n
                    ring.commit(sid, n, final=(offs[i] == d.size))
            recs, fin = ring.poll(sid)                       # no call fails because of what the bytes are
            got[i].append(recs.copy())
            done[i] = fin
        ring.pump()
    assert all(done), ring.debug()
    for i, d in enumerate(datas):
        _assert_same(np.concatenate(got[i]), O.chunk_and_digest(cfg, d, [(0, d.size)]), i)
        ring.close_stream(sids[i])
    ring.quiesce()
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    ring.close()
    eng.close()


@pytest.mark.parametrize("extra", [[], ["--ring-kind", "0", "--ring-streams", "2"]])
def test_bench_ring_workload_small_scale(gpu_lib, extra):
    """bench.py's default workload (the ring) end to end at a size the oracle re-checks completely enough: 1.5 GiB files
    through a 6 GiB arena, prefix + restart points over every timed file."""
    out = subprocess.run([sys.executable, os.path.join(ROOT, "bench.py"), "--gib", "1.5", "--arena-gib", "6", "--steps", "5",
                          "--warmup", "2", "--cpu-sample-gib", "0.5", "--no-extras", "--ring-sha-cus", "128"] + extra,
                         capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stdout[-1000:] + out.stderr[-3000:]
    d = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith("{")][0])
    c = d["cpu_baseline"]
    assert c["records_match_gpu"] is True and c["whole_file_restart_points"]["files"] == 6, c
    assert c["whole_file_restart_points"]["max_offset"] == d["config"]["bytes_per_step"]
    assert d["roofline"]["service_launch_bytes"] == 5 * d["config"]["bytes_per_step"]
    assert d["value"] > 2 and d["config"]["distinct_data_per_step"] is True   # (7.5 GiB + a 0.6 s drain: no throughput claim here)


def test_cpp_mirror_page_ring(gpu_lib, O, tmp_path):
    """include/pbsgpu.hpp's transfer::PageRing driven by a C++ program (tests/native/test_cpp_ring.cpp): three streams,
    entries delivered through the sink, compared with the oracle."""
    exe = str(tmp_path / "test_cpp_ring")
    libdir = os.path.join(ROOT, "pbs_plus_amd", "lib")
    subprocess.run(["g++", "-std=c++17", "-O1", "-Wall", os.path.join(ROOT, "tests", "native", "test_cpp_ring.cpp"),
                    "-L" + libdir, "-lpbsgpu", "-Wl,-rpath," + libdir, "-o", exe], check=True)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "cpp-ring-ok" in out.stdout, out.stdout[-500:] + out.stderr[-1000:]
    jobs = [(71, 0, 5 * 262144 + 777), (72, 3, 9 * 262144), (73, 4, 100)]
    want = _oracle_records(O, 65536, jobs)
    got = {0: [], 1: [], 2: []}
    for ln in out.stdout.splitlines():
        if ln.startswith("C "):
            _, j, end, size, dig = ln.split()
            got[int(j)].append((int(end), int(size), dig))
    for j in range(3):
        assert got[j] == [(int(r["end"]), int(r["size"]), bytes(r["digest"]).hex()) for r in want[j]], j


@pytest.mark.parametrize("seed", range(6))
def test_ring_randomized_programs(gpu_lib, O, seed, monkeypatch):
    """Random programs over the ring: 10-40 streams of random lengths (empty, 1 byte, around the window, around one and
    several pages, up to a few MiB) and data kinds through a SMALL arena (heavy page recycling, streams sharing rounds),
    1-6 streams fed at a time, with and without the long-chunk queue, with and without a tight backlog gate — every stream
    bit-exact vs the oracle."""
    from pbs_plus_amd import PageRing

    rng = np.random.default_rng(1000 + seed)
    avg = [4096, 65536][seed % 2]
    page = 65536 if avg == 4096 else 262144
    npages = int(rng.integers(12, 60))
    if seed % 3 == 0:
        monkeypatch.setenv("PBSGPU_RING_LONG_BYTES", str(avg * 2))
    if seed % 2 == 1:
        # the backlog gate (pages are refused while more than this waits in front of the service) at a limit these small
        # rings reach all the time: a few pages' worth
        monkeypatch.setenv("PBSGPU_RING_BACKLOG_MIB", str(round(float(rng.integers(2, 9)) * page / 1048576.0, 4)))
    eng = _engine(avg)
    ring = PageRing(eng, arena_bytes=npages * (page + 256), page_bytes=page, max_streams=16, sha_cus=int(rng.integers(2, 24)),
                    round_pages=int(rng.integers(2, 12)))
    special = [0, 1, 63, 64, 65, page - 1, page, page + 1, 3 * page, 3 * page + 5, avg * 4, avg * 4 + 1]
    jobs = []
    for i in range(int(rng.integers(10, 40))):
        n = int(rng.choice(special)) if rng.random() < 0.4 else int(rng.integers(1, 12 * page))
        jobs.append((5000 + 100 * seed + i, int(rng.integers(0, 5)), n))
    got = ring.ingest_synthetic(jobs, timeout_s=90.0, concurrent=int(rng.integers(1, 7)))
    ring.quiesce()
    st = ring.stats()
    want = _oracle_records(O, avg, jobs)
    for i, (g, w) in enumerate(zip(got, want)):
        _assert_same(g, w, (seed, i, jobs[i]))
    assert st["pages_free"] == st["pages_total"] and st["pages_recycled"] == st["pages_enqueued"], st
    ring.close()
    eng.close()


def test_ring_survives_a_silent_host(gpu_lib, O, monkeypatch):
    """The service is a kernel that only ends on request, and the goroutine that drives a ring may sit in a blocking tape
    read for as long as it likes (internal/tapeio/converter.go:672-680). A host that keeps calling while it has nothing
    to feed keeps the service; a host that does not call AT ALL for PBSGPU_RING_IDLE_TIMEOUT_S finds the service gone —
    it stopped on its own after a handshake that guarantees no chunk is left behind — and the ring healthy: the next pump
    starts the service again and the stream finishes bit-exact. Twice in a row, once with pages of the stream waiting in
    an open chunk across the silence, and parking by hand in between."""
    import time

    from pbs_plus_amd import PageRing

    monkeypatch.setenv("PBSGPU_RING_IDLE_TIMEOUT_S", "1")
    eng = _engine(4096)
    ring = PageRing(eng, arena_bytes=24 * (65536 + 256), **SMALL)
    sid = ring.open()
    assert ring.fill(sid, 5, 0, 3 * 65536, final=False) == 3 * 65536
    got = []
    t0 = time.time()
    while time.time() - t0 < 3.5:          # 3.5 s with nothing to feed, but the host keeps polling: the service stays
        ring.pump()
        got.append(ring.poll(sid)[0].copy())
        time.sleep(0.01)
    assert ring.stats()["service_launches"] == 1
    total = 3 * 65536
    for rep in range(2):
        time.sleep(4.0)                     # silence: no ring call at all (the service stops after 1-2 timeouts)
        n = 2 * 65536
        assert ring.fill(sid, 5, 0, n, final=False) == n
        total += n
        t0 = time.time()
        while time.time() - t0 < 2.0:
            ring.pump()
            got.append(ring.poll(sid)[0].copy())
            time.sleep(0.005)
        assert ring.stats()["service_launches"] == 2 + rep, ring.stats()
    ring.park()                             # by hand, without waiting; the next round starts a new service behind the old one
    assert ring.fill(sid, 5, 0, 65536 + 17, final=True) == 65536 + 17
    total += 65536 + 17
    t0 = time.time()
    fin = False
    while not fin and time.time() - t0 < 20:
        ring.pump()
        recs, fin = ring.poll(sid)
        got.append(recs.copy())
    assert fin
    want = O.chunk_and_digest(O.new_config(4096), O.fill(total, 5, 0), [(0, total)])
    _assert_same(np.concatenate(got), want, "silent host")
    assert ring.stats()["service_launches"] == 4
    ring.close_stream(sid)
    ring.quiesce()
    st = ring.stats()
    assert st["pages_free"] == st["pages_total"], st
    ring.close()
    eng.close()


def test_ring_create_is_ordered_against_a_busy_null_stream(gpu_lib, O):
    """The ring's device-side state (queue control words, stream states, page reference counts) is cleared at create — on
    the ring's own control stream. A plain hipMemset is queued on the NULL stream and returns at once; the ring's streams
    are non-blocking, i.e. not ordered against it, and the clear used to land after the first rounds once in a few
    hundred creates (scripts/r3_probe_memset_order.py, profiles/r03_probe_hipmemset_ordering.log: 1 in 500 on an idle null
    stream, always on a busy one) — the queue tail and the page counts were wiped, no page ever came back. Here: creates
    right behind 0.1 s of large memsets queued on the null stream through the same HIP runtime, every page must come back.
    (Not a deterministic reproducer of the old defect — create's later allocations happened to wait for the null stream.)"""
    import ctypes as C

    from pbs_plus_amd import PageRing

    hip = C.CDLL("libamdhip64.so.7", mode=os.RTLD_NOLOAD)      # the runtime instance libpbsgpu.so is linked against
    hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
    hip.hipMemset.restype = C.c_int
    jobs = [(91, 0, (1 << 20) + 5), (92, 1, 300 * 1024), (93, 4, 700 * 1024 + 3), (94, 0, 64)]
    want = _oracle_records(O, 4096, jobs)
    eng = _engine(4096)
    big = eng.alloc(64 << 30)
    for rep in range(4):
        for _ in range(8):
            assert hip.hipMemset(big.ptr, rep, big.nbytes) == 0     # asynchronous for device memory: queued on the null stream
        ring = PageRing(eng, arena_bytes=10 * (65536 + 256), page_bytes=65536, max_streams=2, sha_cus=2, round_pages=3)
        got = ring.ingest_synthetic(jobs, timeout_s=30.0, concurrent=2)
        ring.quiesce()
        for i, (g, w) in enumerate(zip(got, want)):
            _assert_same(g, w, (rep, i))
        st = ring.stats()
        assert st["pages_free"] == st["pages_total"], st
        ring.close()
    big.free()
    eng.close()
