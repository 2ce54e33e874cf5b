#!/usr/bin/env python3
"""Host-fed throughput of the streaming writer (PCIe-inclusive): bytes written from host memory
through PayloadStream -> records. Not the headline metric (bench.py is device-resident)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, PayloadStream, buzhash

total_gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16
window_gib = float(sys.argv[2]) if len(sys.argv) > 2 else 1
inflight = int(sys.argv[3]) if len(sys.argv) > 3 else 4
eng = Engine(buzhash.NewConfig(4 << 20), inflight=inflight)
rng = np.random.default_rng(1)
piece = rng.integers(0, 256, 256 << 20, dtype=np.uint8)
ps = PayloadStream(eng, window_bytes=int(window_gib * (1 << 30)))
n = int(total_gib * 4)
t0 = time.perf_counter()
nrec = 0
zero_copy = len(sys.argv) > 4 and sys.argv[4] == "zc"
for i in range(n):
    if zero_copy:  # fill the library's pinned staging directly (what io.ReadFull would do)
        off = 0
        while off < piece.size:
            buf = ps.reserve()
            k = min(buf.size, piece.size - off)
            buf[:k] = piece[off:off + k]
            ps.commit(k)
            off += k
    else:
        ps.write(piece)
    nrec += ps.poll().size
ps.finish()
nrec += ps.poll().size
dt = time.perf_counter() - t0
print(f"stream{' zero-copy' if zero_copy else ''}: {total_gib} GiB, window {window_gib} GiB, inflight {inflight}: {total_gib/dt:.2f} GiB/s, {nrec} records, {dt:.2f} s")
