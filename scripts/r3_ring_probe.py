#!/usr/bin/env python3
"""Round-3 probe of the page ring (pbsgpu_ring_*): parity scenarios of growing size against the oracle, then a
throughput sweep. One JSON line per scenario; every scenario is fenced (exceptions, timeouts) so that one GPU call
reports as much as possible. Usage: python scripts/r3_ring_probe.py [parity] [perf] [--gib N]"""
import json
import os
import sys
import threading
import time
import traceback

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("GPU_MAX_HW_QUEUES", "24")
os.environ.setdefault("PBSGPU_RING_IDLE_TIMEOUT_S", "4")

import numpy as np  # noqa: E402

GiB = 1 << 30


def emit(**kw):
    print(json.dumps(kw), flush=True)


def compare(O, avg, jobs, got):
    cfg = O.new_config(avg)
    out = []
    lock = threading.Lock()

    def one(i):
        seed, kind, n = jobs[i]
        want = O.chunk_and_digest(cfg, O.fill(n, seed, kind), [(0, n)]) if n else np.zeros(0, dtype=O.RECORD_DTYPE)
        g = got[i]
        ok = (g.size == want.size and np.array_equal(g["end"], want["end"]) and np.array_equal(g["digest"], want["digest"])
              and np.array_equal(g["size"], want["size"]))
        info = {"job": i, "bytes": n, "kind": kind, "ok": bool(ok), "gpu": int(g.size), "oracle": int(want.size)}
        if not ok:
            m = min(g.size, want.size)
            bad = np.flatnonzero((g["end"][:m] != want["end"][:m]) | np.any(g["digest"][:m] != want["digest"][:m], axis=1))
            info["first_bad"] = int(bad[0]) if bad.size else m
            j = info["first_bad"]
            info["gpu_ends"] = [int(x) for x in g["end"][max(0, j - 1):j + 3]]
            info["want_ends"] = [int(x) for x in want["end"][max(0, j - 1):j + 3]]
            if j < m:
                info["end_same"] = bool(g["end"][j] == want["end"][j])
        with lock:
            out.append(info)

    ths = [threading.Thread(target=one, args=(i,)) for i in range(len(jobs))]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    return sorted(out, key=lambda x: x["job"])


def parity():
    import pbs_plus_amd
    from oracle import oracle as O
    from pbs_plus_amd import buzhash

    O.build()
    scen = [
        ("A one small stream", 4096, dict(arena_bytes=24 * (65536 + 256), page_bytes=65536, max_streams=4, sha_cus=4, round_pages=4),
         [(11, 0, 200 * 1024 + 17)], None),
        ("B mixed streams, empty and tiny", 4096, dict(arena_bytes=24 * (65536 + 256), page_bytes=65536, max_streams=8, sha_cus=4, round_pages=6),
         [(21, 0, (1 << 20) + 5), (22, 1, 300 * 1024), (23, 3, 700 * 1024 + 3), (24, 0, 0), (25, 0, 63), (26, 2, 65536), (27, 0, 65536 * 3)], None),
        ("B2 same, two streams at a time", 4096, dict(arena_bytes=10 * (65536 + 256), page_bytes=65536, max_streams=2, sha_cus=2, round_pages=3),
         [(31, 0, (1 << 20) + 5), (32, 1, 300 * 1024), (33, 3, 700 * 1024 + 3), (34, 0, 64), (35, 0, 65), (36, 0, 131072)], 2),
        ("C avg 64 KiB", 65536, dict(arena_bytes=96 * (262144 + 256), page_bytes=262144, max_streams=8, sha_cus=16, round_pages=16),
         [(41 + i, i % 4, (8 << 20) + 4099 * i) for i in range(6)], None),
        ("D production avg, 3 x 1.5 GiB", 4 << 20, dict(arena_bytes=5 * GiB, max_streams=4, sha_cus=64, round_pages=64),
         [(51, 0, 3 * GiB // 2 + 8 * 7), (52, 3, 3 * GiB // 2), (53, 1, GiB // 2 + 4096)], None),
    ]
    for name, avg, opt, jobs, conc in scen:
        t0 = time.perf_counter()
        try:
            eng = pbs_plus_amd.Engine(buzhash.NewConfig(avg), device=0, inflight=1)
            ring = pbs_plus_amd.PageRing(eng, **opt)
            got = ring.ingest_synthetic(jobs, timeout_s=60.0, concurrent=conc)
            ring.quiesce()
            st = ring.stats()
            res = compare(O, avg, jobs, got)
            emit(scenario=name, ok=all(r["ok"] for r in res), seconds=round(time.perf_counter() - t0, 2),
                 details=[r for r in res if not r["ok"]][:4], records=sum(r["gpu"] for r in res),
                 stats={k: st[k] for k in ("pages_total", "pages_free", "rounds", "chunks", "pages_enqueued", "pages_recycled",
                                           "service_launches", "service_ms_last")})
            ring.close()
            eng.close()
        except BaseException as exc:  # noqa: BLE001
            emit(scenario=name, ok=False, error=repr(exc), trace=traceback.format_exc()[-1500:],
                 seconds=round(time.perf_counter() - t0, 2))
            try:
                emit(scenario=name, stats_after_error=ring.stats())
            except Exception:
                pass


def perf(total_gib, stream_gib=32.0, nstreams=8, arena_gib=232.0, knobs=()):
    import pbs_plus_amd
    from pbs_plus_amd import buzhash

    eng = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=1)
    for kn in knobs or ({},):
        t0 = time.perf_counter()
        try:
            ring = pbs_plus_amd.PageRing(eng, arena_bytes=int(arena_gib * GiB), max_streams=max(64, nstreams),
                                         sha_cus=kn.get("sha_cus", 0), round_pages=kn.get("round_pages", 0))
            per = int(stream_gib * GiB)
            njobs = max(nstreams, int(total_gib / stream_gib))
            jobs = [(1000 + i, 0, per) for i in range(njobs)]
            tb = time.perf_counter()
            got = ring.ingest_synthetic(jobs, timeout_s=180.0, concurrent=nstreams)
            ring.quiesce()
            dt = time.perf_counter() - tb
            st = ring.stats()
            nrec = sum(g.size for g in got)
            nbytes = sum(int(g["size"].sum()) for g in got)
            emit(scenario="perf", knobs=kn, streams=nstreams, stream_gib=stream_gib, GiBps=round(nbytes / GiB / dt, 1),
                 seconds=round(dt, 3), records=nrec, bytes=nbytes, bytes_ok=(nbytes == njobs * per),
                 service_ms=st["service_ms_last"], rounds=st["rounds"], pages_total=st["pages_total"],
                 sha_cus=st["sha_cus"], setup_s=round(tb - t0, 2))
            ring.close()
        except BaseException as exc:  # noqa: BLE001
            emit(scenario="perf", knobs=kn, ok=False, error=repr(exc), trace=traceback.format_exc()[-1200:])
            try:
                emit(scenario="perf", stats_after_error=ring.stats())
                ring.close()
            except Exception:
                pass
    eng.close()


if __name__ == "__main__":
    args = sys.argv[1:]
    if "parity" in args or not args:
        parity()
    if "perf" in args:
        total = 512.0
        if "--gib" in args:
            total = float(args[args.index("--gib") + 1])
        perf(total, knobs=[{}, {"sha_cus": 192}, {"sha_cus": 224}, {"round_pages": 128}])
