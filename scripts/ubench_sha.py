#!/usr/bin/env python3
"""SHA-256 kernel occupancy sweep: N equal segments (one lane each) -> time per 64-byte block
per lane and aggregate GB/s. Tells how single-wave issue rate and waves/SIMD shape the kernel."""
import os, sys, time, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, buzhash

eng = Engine(buzhash.NewConfig(4 << 20))
seg = int(sys.argv[1]) if len(sys.argv) > 1 else 128 << 10
counts = [64, 1024, 16384, 65536, 131072, 262144]
total = max(counts) * seg
buf = eng.alloc(total)
eng.fill(buf.ptr, total, 1, 0)
res = []
for n in counts:
    segs = np.stack([np.arange(n, dtype=np.uint64) * np.uint64(seg), np.full(n, seg, dtype=np.uint64)], axis=1)
    eng.sha256_many(buf, segs[: min(n, 64)], nbytes=total)  # warm
    t0 = time.perf_counter(); d = eng.sha256_many(buf, segs, nbytes=total); dt = time.perf_counter() - t0
    blocks = seg // 64 + 1
    res.append({"lanes": n, "waves": n // 64, "ms": round(dt * 1e3, 3), "us_per_block_per_lane": round(dt * 1e6 / blocks, 3),
                "GBps": round(n * seg / dt / 1e9, 1)})
    print(res[-1], flush=True)
import hashlib
h = buf.download(0, seg)
assert bytes(d[0]) == hashlib.sha256(h.tobytes()).digest()
print(json.dumps(res))
