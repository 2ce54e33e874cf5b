"""Does a concurrently running scan slow the SHA-256 chain (chip clock under the scan's power draw)?
usage: r2_probe_clock.py   (PBSGPU_SCAN_CU_RESERVE limits the CUs the scan may use)"""
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
t0 = time.perf_counter(); eng.candidates(buf, n); t_scan = time.perf_counter() - t0
t0 = time.perf_counter(); eng.candidates(buf, n); t_scan = min(t_scan, time.perf_counter() - t0)
print("reserve=%s: 16 GiB scan alone %.2f ms (%.2f TB/s); sha alone %.1f ms" % (os.environ.get("PBSGPU_SCAN_CU_RESERVE", "16"), t_scan * 1e3, n / t_scan / 1e12, sha()))
stop = False
cnt = [0]
def scans(sz):
    while not stop:
        eng.candidates(buf, sz); cnt[0] += 1
th = threading.Thread(target=scans, args=(n,)); th.start(); time.sleep(0.05)
t = [sha() for _ in range(2)]
stop = True; th.join()
print("   sha with back-to-back 16 GiB scans: %s ms (%d scans)" % (["%.1f" % x for x in t], cnt[0]))
