#!/usr/bin/env python3
"""Is a plain hipMemset (null stream) ordered against work on the library's non-blocking streams, and when does it run?
For each iteration: hipMemset(buf, 0x55) on the null stream, then the engine's generator writes buf on a non-blocking
stream and waits for it, then (after a pause) buf is read back: 0x55 bytes in the result = the memset ran AFTER the
generator although it was issued before it."""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

eng = pbs_plus_amd.Engine(buzhash.NewConfig(4096), device=0, inflight=1)
hip = C.CDLL("libamdhip64.so.7", mode=os.RTLD_NOLOAD)
hip.hipMemset.argtypes = [C.c_void_p, C.c_int, C.c_size_t]
hip.hipMemset.restype = C.c_int
n = 1 << 16
buf = eng.alloc(n)
big = eng.alloc(32 << 30)
iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
for label, busy in (("idle null stream", 0), ("null stream busy with 4 x 32 GiB memsets", 4)):
    late = 0
    t_call = 0.0
    for it in range(iters):
        for _ in range(busy):
            assert hip.hipMemset(big.ptr, 1, big.nbytes) == 0
        t0 = time.perf_counter()
        assert hip.hipMemset(buf.ptr, 0x55, n) == 0
        t_call += time.perf_counter() - t0
        eng.fill(buf.ptr, n, seed=it + 1, kind=0)
        if it % 50 == 0:
            time.sleep(0.05)
        got = buf.download()
        late += int(np.count_nonzero(got == 0x55) > n // 2)
    print(f"{label}: {late} of {iters} memsets landed after the generator; hipMemset call {t_call / iters * 1e6:.1f} us on average", flush=True)
    iters = max(20, iters // 10)
