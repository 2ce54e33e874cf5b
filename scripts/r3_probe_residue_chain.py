#!/usr/bin/env python3
"""Residue behind host-fed volume, narrowed down: the serial SHA-256 chain of one 64 GiB file (avg 4 MiB, one slot, nothing
else running) over (a) a buffer allocated BEFORE eight streams wrote 4 GiB each and (b) a buffer allocated AFTER, in the
same process. Slower on both = not the placement of the buffer; slower only on (b) = placement (page-table fragments)."""
import os
import sys
import threading
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

n = 64 << 30
eng = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=1)


def chain(buf, label):
    best = None
    for _ in range(2):
        tk = eng.submit(buf, [(0, n)], nbytes=n)
        eng.wait(tk)
        tm = eng.timing(tk)
        eng.collect(tk)
        if best is None or tm["sha_ms"] < best["sha_ms"]:
            best = tm
    print(f"{label}: sha {best['sha_ms']:.1f} ms, scan {best['scan_ms']:.2f} ms", flush=True)


def writers(P, mib):
    e2 = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=2)
    src = np.random.default_rng(3).integers(0, 256, 32 << 20, dtype=np.uint8)

    def one(i):
        st = pbs_plus_amd.PayloadStream(e2, 256 << 20)
        for _ in range(max(1, mib // 32)):
            st.write(src)
        st.finish()
        st.poll()
        st.close()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(P)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    e2.close()


old = eng.alloc(n)
eng.fill(old.ptr, n, seed=7, kind=0)
chain(old, "buffer A, fresh process")
for rnd, mib in enumerate([int(x) for x in os.environ.get("PROBE_MIB", "4096,16384").split(",")]):
    t0 = time.perf_counter()
    writers(8, mib)
    print(f"-- eight streams wrote {mib} MiB each in {time.perf_counter() - t0:.1f} s, everything closed --", flush=True)
    chain(old, "buffer A (allocated before)")
    new = eng.alloc(n)
    eng.fill(new.ptr, n, seed=7, kind=0)
    chain(new, "buffer B (allocated after)")
    new.free()
