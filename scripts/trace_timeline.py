#!/usr/bin/env python3
"""Print a compact per-launch timeline (queue, kernel, start, duration) from a rocprofv3 kernel trace CSV."""
import csv, sys
rows = list(csv.DictReader(open(sys.argv[1])))
t0 = min(int(r["Start_Timestamp"]) for r in rows)
keep = ("k_scan", "k_sha256", "k_resolve<true", "k_order")
for r in sorted(rows, key=lambda r: int(r["Start_Timestamp"])):
    n = r["Kernel_Name"]
    if not any(k in n for k in keep):
        continue
    short = "scan" if ("k_scan<" in n or "k_scan2<" in n or "k_scan3<" in n) else "scanaux" if "k_scan" in n else "sha" if "k_sha256" in n else "order" if "k_order" in n else "resolveW"
    s, e = int(r["Start_Timestamp"]) - t0, int(r["End_Timestamp"]) - t0
    print(f"q={r.get('Queue_Id','?'):>3} {short:8s} start={s/1e6:9.2f} ms dur={(e-s)/1e6:8.2f} ms")
