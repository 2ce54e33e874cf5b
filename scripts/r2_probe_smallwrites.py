"""Small writes (what io.Copy does): 32 KiB and 1 MiB write() calls through one stream."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, PayloadStream, buzhash
eng = Engine(buzhash.NewConfig(4 << 20), 0, 2)
buf = eng.alloc(1 << 30); eng.fill(buf.ptr, 1 << 30, 3, 0); src = buf.download(); buf.free()
for wsz in (32 << 10, 1 << 20, 32 << 20):
    st = PayloadStream(eng, 256 << 20)
    for off in range(0, 1 << 30, 32 << 20): st.write(src[off:off + (32 << 20)])   # warm the ring
    t0 = time.perf_counter()
    total = 4 << 30
    mv = memoryview(src)
    L = st._L; h = st._h; base = src.ctypes.data
    for off in range(0, total, wsz):
        L.pbsgpu_stream_write(h, base + (off % (1 << 30)), wsz)
    dt = time.perf_counter() - t0
    st.finish(); st.poll(); st.close()
    print("write size %8d: %.2f GiB/s (%.2f us per call)" % (wsz, total / dt / 2**30, dt / (total / wsz) * 1e6), flush=True)
