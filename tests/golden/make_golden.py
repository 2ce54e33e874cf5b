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
    """segs: (seed, kind, length) generator segments, or (pattern bytes, length): a crafted period repeated (round 6)"""
    cfg = O.new_config(avg)
    parts, table, off, spec = [], [], 0, []
    for sg in segs:
        if len(sg) == 2:
            pat, n = sg
            parts.append(np.tile(pat, n // pat.size + 1)[:n].copy())
            spec.append({"kind": 100, "pattern": bytes(pat).hex(), "length": n})
        else:
            seed, kind, n = sg
            parts.append(O.fill(n, seed, kind))
            spec.append({"seed": seed, "kind": kind, "length": n})
        table.append((off, n))
        off += n
    data = np.concatenate(parts) if parts else np.zeros(0, np.uint8)
    recs = O.chunk_and_digest(cfg, data, table, impl=0)
    return {
        "name": name, "avg": avg, "segments": spec,
        "records": [[int(r["segment"]), int(r["end"]), int(r["size"]), bytes(r["digest"]).hex()] for r in recs],
    }


def crafted_cases():
    """Candidate-DENSE streams (tests/dense_inputs.py): a 64-byte period whose window hash is 0xFFFFFFFF makes every position
    a candidate (the serial chunker cuts at the minimum every time), one that passes the break test at one phase gives a
    candidate per period, a period of 16 cancels to none (max-size cuts). The engine resolves such data exactly since round 6
    (rounds 3-5: PBSGPU_E_DENSITY on the streaming paths); a maintainer with the Go module pins the same behaviour with
    tools/golden/main.go, which reads the patterns from THIS file's cases."""
    sys.path.insert(0, os.path.join(ROOT, "tests"))
    import dense_inputs as D

    allp = D.all_candidate_pattern(O.default_table())
    one4k = D.one_phase_pattern(O, O.new_config(4096))
    one64k = D.one_phase_pattern(O, O.new_config(65536))
    assert one4k is not None and one64k is not None
    return [
        ("period64_every_position_avg4k", 4096, [(allp, 300_000)]),
        ("period64_one_phase_avg4k", 4096, [(71, 0, 70_001), (one4k, 200_000), (72, 0, 5_003), (allp, 64), (allp, 65), (allp, 1_024)]),
        ("period64_one_phase_avg64k", 65536, [(one64k, 3 << 20), (73, 3, 1 << 20)]),
        ("period16_no_candidate_avg4k", 4096, [(allp[:16], 100_000)]),
        ("period64_every_position_avg4m", 4 << 20, [(allp, (40 << 20) + 7)]),
    ]


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
           "cases": [build_case(*c) for c in CASES] + [counter_case()] + [build_case(*c) for c in crafted_cases()]}
    path = os.path.join(os.path.dirname(os.path.abspath(__file__)), "chunks_v1.json")
    with open(path, "w") as f:
        json.dump(out, f, indent=0, separators=(",", ":"))
    print(path, os.path.getsize(path), "bytes,", sum(len(c["records"]) for c in out["cases"]), "records")
