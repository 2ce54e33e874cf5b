#!/usr/bin/env python3
"""Repeat the page-starved small-ring scenarios many times (a hang seen once in ~8 runs of the GPU suite): on a timeout
print the ring's device-side state. Usage: python scripts/r3_ring_stress.py [iterations]"""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ.setdefault("PBSGPU_RING_IDLE_TIMEOUT_S", os.environ.get("STRESS_IDLE_S", "5"))
import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 200
jobs = [(31, 0, (1 << 20) + 5), (32, 1, 300 * 1024), (33, 3, 700 * 1024 + 3), (34, 0, 64), (35, 0, 65), (36, 4, 131072)]
ref = None
t0 = time.time()
bad = 0
for it in range(iters):
    eng = pbs_plus_amd.Engine(buzhash.NewConfig(4096), device=0, inflight=1)
    ring = pbs_plus_amd.PageRing(eng, arena_bytes=10 * (65536 + 256), page_bytes=65536, max_streams=2, sha_cus=2, round_pages=3)
    try:
        got = ring.ingest_synthetic(jobs, timeout_s=8.0, concurrent=2)
        ring.quiesce()
        sig = [(int(g.size), bytes(g["digest"].tobytes()[:64])) for g in got]
        if ref is None:
            ref = sig
        elif sig != ref:
            print("iteration", it, "DIFFERENT RESULT", flush=True)
            bad += 1
    except TimeoutError as exc:
        print("iteration", it, "HANG", str(exc)[:6000], flush=True)
        bad += 1
        try:
            ring.quiesce()
        except Exception as e2:  # noqa: BLE001
            print("quiesce after hang:", e2)
    ring.close()
    eng.close()
    if bad >= 3:
        break
print(f"stress: {it + 1} iterations, {bad} bad, {time.time() - t0:.1f} s")
