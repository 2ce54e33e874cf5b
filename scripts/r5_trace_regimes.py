#!/usr/bin/env python3
"""Cut-side anatomy of the page ring from a rocprofv3 kernel trace of `bench.py` (CSV from scripts/rocpd_export.py, .gz ok).

For the longest service launch (= the timed region of the driver's command) it prints, over a steady-state window:
  * the share of the wall time each cut-side kernel is running, and how much of it pairs of kernels OVERLAP;
  * rounds in the window, mean / median duration per kernel and per round;
  * the per-100 ms round counts from the start of the launch (ramp-up: full rounds; steady state: one gate quantum each).
Usage: r5_trace_regimes.py trace.csv[.gz] [t0_ms t1_ms]   (window relative to the service launch, default 400..1600)
"""
import collections
import csv
import gzip
import io
import statistics
import sys


def load(path):
    f = io.TextIOWrapper(gzip.open(path)) if path.endswith(".gz") else open(path)
    rows = list(csv.DictReader(f))
    for r in rows:
        r["s"] = int(r["Start_Timestamp"])
        r["e"] = int(r["End_Timestamp"])
    return rows


def union(ivs):
    out = []
    for a, b in sorted(ivs):
        if out and a <= out[-1][1]:
            out[-1][1] = max(out[-1][1], b)
        else:
            out.append([a, b])
    return out


def length(ivs):
    return sum(b - a for a, b in ivs)


def inter(A, B):
    i = j = 0
    out = []
    while i < len(A) and j < len(B):
        a, b = max(A[i][0], B[j][0]), min(A[i][1], B[j][1])
        if a < b:
            out.append([a, b])
        if A[i][1] < B[j][1]:
            i += 1
        else:
            j += 1
    return out


def main():
    rows = load(sys.argv[1])
    w0, w1 = (float(sys.argv[2]), float(sys.argv[3])) if len(sys.argv) > 3 else (400.0, 1600.0)
    svc = max((r for r in rows if "k_sha256_pair" in r["Kernel_Name"]), key=lambda r: r["e"] - r["s"])
    print(f"service launch {(svc['e'] - svc['s']) / 1e6:.1f} ms; window {w0:.0f}..{w1:.0f} ms of it")
    t0, t1 = svc["s"] + w0 * 1e6, svc["s"] + w1 * 1e6
    names = {"scan": "k_scan3", "refill": "k_ring_fill", "control": "k_ring_control", "prep": "k_ring_prep(", "stage": "k_ring_stage"}
    iv = {}
    for k, n in names.items():
        iv[k] = union([(max(r["s"], t0), min(r["e"], t1)) for r in rows if n in r["Kernel_Name"] and r["e"] > t0 and r["s"] < t1])
    W = t1 - t0
    print("share of the wall time:", {k: round(length(v) / W, 3) for k, v in iv.items()},
          "any:", round(length(union(sum((list(map(tuple, v)) for v in iv.values()), []))) / W, 3))
    print("overlap:", {f"{a}&{b}": round(length(inter(iv[a], iv[b])) / W, 4) for a, b in (("scan", "refill"), ("scan", "control"), ("refill", "control"))})
    nround = 0
    for k, n in names.items():
        d = [(r["e"] - r["s"]) / 1e3 for r in rows if n in r["Kernel_Name"] and r["s"] > t0 and r["e"] < t1]
        if d:
            nround = max(nround, len(d))
            print(f"  {k:8s} launches {len(d):5d}  mean {statistics.mean(d):8.1f} us  median {statistics.median(d):8.1f}  p90 {sorted(d)[9 * len(d) // 10]:8.1f}")
    if nround:
        print(f"rounds in the window {nround}: one per {W / 1e3 / nround:.0f} us")
    bins = collections.Counter()
    for r in rows:
        if names["control"] in r["Kernel_Name"] and svc["s"] - 50e6 < r["s"] < svc["e"]:
            bins[int((r["s"] - svc["s"]) // 100e6)] += 1
    print("control launches per 100 ms from the launch:", [bins[k] for k in sorted(bins)])


if __name__ == "__main__":
    main()
