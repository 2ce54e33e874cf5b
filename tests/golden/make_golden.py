#!/usr/bin/env python3
"""Generates tests/golden/chunks_v1.json from the CPU oracle (oracle/).

PARITY UNPINNED against github.com/pbs-plus/pxar v0.34.0: these vectors pin the
*restatement* (published Proxmox/casync chunker + FIPS 180-4 SHA-256), so that the oracle,
the HIP engine and any future Go-module vectors can be diffed against one committed file.
tools/golden/main.go emits the same JSON schema from the real Go module for a maintainer.

Run from the repo root:  python tests/golden/make_golden.py
"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import oracle as O  # noqa: E402

CASES = [
    # name, avg, [(seed, kind, length) per segment]
    ("rand_avg4k", 4096, [(11, 0, 1 << 20)]),
    ("mixed_avg4k", 4096, [(21, 0, 300_000), (22, 1, 70_000), (23, 2, 200_001), (24, 3, 400_003), (25, 0, 17), (26, 0, 0),
                           (27, 0, 1024), (28, 0, 1089)]),
    ("rand_avg64k", 65536, [(31, 0, 8 << 20)]),
    ("zero_extents_avg64k", 65536, [(41, 3, 6 << 20), (42, 1, 1 << 20)]),
    ("rand_avg4m", 4 << 20, [(51, 0, 48 << 20)]),
    # generator kind 4 (two ChaCha quarter-rounds per 16 bytes): the bytes the page-ring bench refills its pages with
    ("arx_avg4k", 4096, [(61, 4, 1 << 20), (62, 4, 65)]),
    ("arx_avg4m", 4 << 20, [(63, 4, 40 << 20)]),
    # upstream Proxmox chunker test input: LE u32 counters 0..262143, avg 64 KiB (generated, not filled)
]


def build_case(name, avg, segs):
    cfg = O.new_config(avg)
    parts, table, off = [], [], 0
    for seed, kind, n in segs:
        parts.append(O.fill(n, seed, kind))
        table.append((off, n))
        off += n
    data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    recs = O.chunk_and_digest(cfg, data, table, impl=0)
    return {
        "name": name, "avg": avg,
        "segments": [{"seed": s, "kind": k, "length": n} for s, k, n in segs],
        "records": [[int(r["segment"]), int(r["end"]), int(r["size"]), bytes(r["digest"]).hex()] for r in recs],
    }


def counter_case():
    buf = np.arange(256 * 1024, dtype="<u4").view(np.uint8)
    cfg = O.new_config(64 * 1024)
    recs = O.chunk_and_digest(cfg, buf, [(0, buf.size)], impl=0)
    return {
        "name": "le_u32_counter_avg64k", "avg": 65536, "segments": "le_u32_counter_262144",
        "records": [[int(r["segment"]), int(r["end"]), int(r["size"]), bytes(r["digest"]).hex()] for r in recs],
    }


if __name__ == "__main__":
    out = {"schema": "pbsgpu-golden-v1", "generator": "oracle (C restatement), parity unpinned vs pbs-plus/pxar v0.34.0",
           "cases": [build_case(*c) for c in CASES] + [counter_case()]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chunks_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes,", sum(len(c["records"]) for c in out["cases"]), "records")
