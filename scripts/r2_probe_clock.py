"""Does a concurrently running scan slow the SHA-256 chain (chip clock under the scan's power draw)?"""
import os, sys, time, threading
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
from pbs_plus_amd import Engine, buzhash
eng = Engine(buzhash.NewConfig(4 << 20), 0, 2)
n = 16 << 30
buf = eng.alloc(n)
eng.fill(buf.ptr, n, 1, 0)
segs = [(i * (16 << 20), 16 << 20) for i in range(64)]
def sha():
    t0 = time.perf_counter(); eng.sha256_many(buf, segs); return (time.perf_counter() - t0) * 1e3
print("sha alone: %.1f ms" % sha(), "%.1f ms" % sha())
stop = False
cnt = [0]
def scans(sz):
    while not stop:
        eng.candidates(buf, sz); cnt[0] += 1
for sz, label in ((n, "16 GiB scans back to back"), (n // 16, "1 GiB scans back to back")):
    stop = False; cnt[0] = 0
    th = threading.Thread(target=scans, args=(sz,)); th.start(); time.sleep(0.05)
    t = [sha() for _ in range(3)]
    stop = True; th.join()
    print("sha with %s: %s ms (%d scans ran)" % (label, ["%.1f" % x for x in t], cnt[0]))
print("sha alone again: %.1f ms" % sha())
