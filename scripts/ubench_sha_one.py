#!/usr/bin/env python3
"""One SHA-256 launch at a given lane count (for rocprofv3 counter passes)."""
import os, sys, time
import numpy as np
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from pbs_plus_amd import Engine, buzhash
n = int(sys.argv[1]); seg = int(sys.argv[2]) if len(sys.argv) > 2 else 131136
eng = Engine(buzhash.NewConfig(4 << 20))
total = n * seg + 4096
buf = eng.alloc(total); eng.fill(buf.ptr, total & ~7, 1, 0)
segs = [(i * seg, seg) for i in range(n)]
for _ in range(2):
    t0 = time.perf_counter(); eng.sha256_many(buf, segs, nbytes=total); dt = time.perf_counter() - t0
print(f"lanes={n} ms={dt*1e3:.2f} GB/s={n*seg/dt/1e9:.1f}")
