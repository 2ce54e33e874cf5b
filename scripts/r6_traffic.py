#!/usr/bin/env python3
"""profiles/r06_traffic.json from the FETCH_SIZE / WRITE_SIZE passes of scripts/r6_final.sh on scripts/r4_ring_pmc.py.

HBM bytes per kernel = launches x mean counter value, corrected as /opt/skills/guides/MI355X_MICROARCH.md prescribes for
gfx950: FETCH_SIZE is in KiB and counts 128-byte requests as 64 B (x 1024 x 2), WRITE_SIZE in KiB (x 1024).
Usage: r5_traffic.py fetch.csv write.csv run.json out.json   (run.json: the JSON line r4_ring_pmc.py printed)"""
import csv
import json
import sys


def table(path):
    t = {}
    for r in csv.DictReader(open(path)):
        t[r["Kernel_Name"]] = float(r["Launches"]) * float(r["Mean"])
    return t


def pick(t, key):
    return sum(v for k, v in t.items() if key in k)


def main():
    fetch, write = table(sys.argv[1]), table(sys.argv[2])
    run = json.loads([l for l in open(sys.argv[3]) if l.startswith("{")][-1])
    alg = int(run["service_bytes"])
    rd = lambda k: pick(fetch, k) * 1024.0 * 2.0
    wr = lambda k: pick(write, k) * 1024.0
    kern = {
        "k_scan3<34,4> (ring rounds, tile->page indirection)": {"hbm_read_bytes": rd("k_scan3"), "hbm_write_bytes": wr("k_scan3")},
        "k_sha256_pair<RingSource,false> + k_sha256_xpair<RingSource> (both services)": {
            "hbm_read_bytes": rd("k_sha256_pair") + rd("k_sha256_xpair"), "hbm_write_bytes": wr("k_sha256_pair") + wr("k_sha256_xpair")},
        "k_ring_fill (synthetic producer, bench only)": {"hbm_read_bytes": rd("k_ring_fill"), "hbm_write_bytes": wr("k_ring_fill")},
        "k_ring_control + k_ring_prep + k_ring_prep_pages + k_ring_stage": {
            "hbm_read_bytes": sum(rd(k) for k in ("k_ring_control", "k_ring_prep", "k_ring_stage")),
            "hbm_write_bytes": sum(wr(k) for k in ("k_ring_control", "k_ring_prep", "k_ring_stage"))},
    }
    for v in kern.values():
        v["read_ratio_vs_algorithmic"] = round(v["hbm_read_bytes"] / alg, 5)
        v["write_ratio_vs_algorithmic"] = round(v["hbm_write_bytes"] / alg, 5)
        v["hbm_read_bytes"], v["hbm_write_bytes"] = int(v["hbm_read_bytes"]), int(v["hbm_write_bytes"])
    total = sum(v["hbm_read_bytes"] + v["hbm_write_bytes"] for v in kern.values())
    names = list(kern)
    out = {
        "source": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes, no other trace domain) on `python3 scripts/r4_ring_pmc.py 48 4` "
                  "(4 streams x 48 GiB through the page ring with PBSGPU_RING_DEFER_SERVICE=1: the cut rounds fill the queue, then both "
                  "services run ALONE as ordinary dispatches the counter passes can measure), round 6 code (round tables staged by k_ring_stage)",
        "corrections": "FETCH_SIZE is reported in KiB and counts 128-byte requests at 64 B on gfx950 (MI355X_MICROARCH.md, HBM section): "
                       "bytes = FETCH_SIZE * 1024 * 2. WRITE_SIZE (KiB) 1:1.",
        "algorithmic_bytes": alg,
        "kernels": kern,
        "ring": {
            "hbm_bytes_per_algorithmic_byte": round(total / alg, 4),
            "note": "MEASURED on the ring's own kernels (round 6): HBM bytes per input byte = refill write %.3f (bench only) + scan read %.3f + "
                    "SHA-256 services read %.3f (+ %.5f everything else); FETCH_SIZE x 1024 x 2, WRITE_SIZE x 1024, "
                    "profiles/r06_pmc_{fetch,write}_size_ring_both_services.csv. The bench line multiplies this ratio by the bytes its service launch hashed."
                    % (kern[names[2]]["write_ratio_vs_algorithmic"], kern[names[0]]["read_ratio_vs_algorithmic"],
                       kern[names[1]]["read_ratio_vs_algorithmic"],
                       total / alg - kern[names[2]]["write_ratio_vs_algorithmic"] - kern[names[0]]["read_ratio_vs_algorithmic"]
                       - kern[names[1]]["read_ratio_vs_algorithmic"]),
        },
    }
    json.dump(out, open(sys.argv[4], "w"), indent=1)
    print(out["ring"]["note"])


if __name__ == "__main__":
    main()
