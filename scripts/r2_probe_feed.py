"""P producer threads, long-lived streams: per-thread write-phase time vs drain time."""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, PayloadStream, buzhash
from oracle import oracle as O
P = int(sys.argv[1]); G = int(sys.argv[2])
eng = Engine(buzhash.NewConfig(4 << 20), 0, 2)
src = [O.fill(1 << 30, 5 + i, 0) for i in range(P)]
gate = threading.Barrier(P + 1)
res = [None] * P
def prod(i):
    st = PayloadStream(eng, 256 << 20)
    for off in range(0, 2 << 30, 32 << 20):
        st.write(src[i][off % (1 << 30):off % (1 << 30) + (32 << 20)])
    gate.wait(); gate.wait()
    t0 = time.perf_counter()
    for g in range(G):
        for off in range(0, 1 << 30, 32 << 20):
            st.write(src[i][off:off + (32 << 20)])
        st.poll(4096)
    t1 = time.perf_counter()
    st.finish()
    t2 = time.perf_counter()
    res[i] = (t1 - t0, t2 - t1)
    st.close()
ths = [threading.Thread(target=prod, args=(i,)) for i in range(P)]
for t in ths: t.start()
gate.wait(); t0 = time.perf_counter(); gate.wait()
for t in ths: t.join()
dt = time.perf_counter() - t0
print("P=%d G=%d: total %.2f s = %.1f GiB/s; write phases %s; drains %s" % (P, G, dt, P * G / dt,
      ["%.2f" % r[0] for r in res], ["%.2f" % r[1] for r in res]), flush=True)
