#!/usr/bin/env python3
"""What does a host-fed leg leave behind that slows LATER work of the same process by 20-30 % (bench.py extras run in a
different order: every leg behind the eight-writer leg lost that much)? Measures one scan + SHA-256 pass over 32 GiB
(avg 64 KiB: throughput-bound) and one ring pass (avg 4 MiB) before and after (a) eight payload streams that each write
a little (all their HIP streams, the copy streams and the hash lanes get used, then everything is closed), (b) the same
with real volume."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

n = 32 << 30


def measure(label):
    eng = pbs_plus_amd.Engine(buzhash.NewConfig(65536), device=0, inflight=1)
    buf = eng.alloc(n)
    eng.fill(buf.ptr, n, seed=7, kind=0)
    best = None
    for _ in range(3):
        t0 = time.perf_counter()
        tk = eng.submit(buf, [(0, n)], nbytes=n)
        eng.wait(tk)
        dt = time.perf_counter() - t0
        tm = eng.timing(tk)
        eng.collect(tk)
        if best is None or dt < best[0]:
            best = (dt, tm)
    buf.free()
    eng.close()
    print(f"{label}: pass {best[0] * 1e3:.1f} ms (scan {best[1]['scan_ms']:.2f}, sha {best[1]['sha_ms']:.2f})", flush=True)


def writers(P, mib_each):
    eng = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=2)
    src = np.random.default_rng(3).integers(0, 256, 32 << 20, dtype=np.uint8)

    def one(i):
        st = pbs_plus_amd.PayloadStream(eng, 256 << 20)
        for _ in range(max(1, mib_each // 32)):
            st.write(src)
        st.finish()
        st.poll()
        st.close()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(P)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    eng.close()


measure("fresh process")
writers(1, 64)
measure("after ONE stream that wrote 64 MiB")
writers(8, 64)
measure("after EIGHT streams that wrote 64 MiB each")
writers(8, 8192)
measure("after EIGHT streams that wrote 8 GiB each")
