"""one producer, trace on: where does a host-fed window spend its time?"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, PayloadStream, buzhash
from oracle import oracle as O
eng = Engine(buzhash.NewConfig(4 << 20), 0, 2)
src = O.fill(1 << 30, 5, 0)
print("h2d GB/s", eng.h2d_bandwidth(1 << 30))
t0 = time.perf_counter(); tmp = src.copy(); print("numpy memcpy 1 GiB: %.1f GB/s" % (src.size / (time.perf_counter() - t0) / 1e9))
st = PayloadStream(eng, 256 << 20)
for rep in range(3):
    t0 = time.perf_counter()
    tw = 0.0
    for off in range(0, src.size, 32 << 20):
        t1 = time.perf_counter()
        st.write(src[off:off + (32 << 20)])
        tw += time.perf_counter() - t1
    dt = time.perf_counter() - t0
    print("rep %d: 1 GiB in %.1f ms (%.2f GiB/s), in write() %.1f ms" % (rep, dt * 1e3, 1 / dt, tw * 1e3), flush=True)
    st.poll()
t0 = time.perf_counter(); st.finish(); print("finish %.1f ms" % ((time.perf_counter() - t0) * 1e3)); print(st.poll().size)
