#!/usr/bin/env python3
"""Does allocation churn (a host-fed stream's window ring: hundreds of 272 MiB buffers allocated on demand and freed at
teardown) slow down LATER large allocations of the same process? Measures one scan + SHA-256 pass over a fresh 32 GiB
buffer (avg 64 KiB chunks: throughput-bound) before and after the churn; stage times from HIP events (pbsgpu_timing)."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

eng = pbs_plus_amd.Engine(buzhash.NewConfig(65536), device=0, inflight=1)
n = 32 << 30
rng = np.random.default_rng(1)


def measure(label):
    buf = eng.alloc(n)
    eng.fill(buf.ptr, n, seed=7, kind=0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        tk = eng.submit(buf, [(0, n)], nbytes=n)
        eng.wait(tk)
        dt = time.perf_counter() - t0
        tm = eng.timing(tk)
        recs = eng.collect(tk)
        if best is None or dt < best[0]:
            best = (dt, tm, recs.size)
    buf.free()
    print(f"{label}: pass {best[0] * 1e3:.1f} ms = {n / 2**30 / best[0]:.0f} GiB/s, stages {best[1]}, {best[2]} chunks", flush=True)


measure("fresh process")
measure("fresh process, second allocation")
for rounds in range(2):
    bufs = [eng.alloc(272 << 20) for _ in range(400)]          # 106 GiB of window-sized buffers
    for i in rng.permutation(len(bufs)):
        bufs[int(i)].free()
    measure(f"after churn round {rounds + 1} (400 x 272 MiB allocated, freed in random order)")
keep = [eng.alloc(272 << 20) for _ in range(400)]
for i in rng.permutation(len(keep))[:200]:
    keep[int(i)].free()
    keep[int(i)] = None
measure("with 200 x 272 MiB scattered buffers still allocated")
eng.close()
