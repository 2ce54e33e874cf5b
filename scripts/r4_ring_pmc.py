"""The page ring's kernels as ORDINARY dispatches, so that rocprofv3's counter passes (which serialise dispatches) can see
them: PBSGPU_RING_DEFER_SERVICE=1 makes the cut rounds only fill the chunk queue; pbsgpu_ring_quiesce then launches the
persistent SHA-256 service ALONE with `stop` already raised — k_sha256_pair<RingSource,false> hashes everything that was
published (every lane busy, chunks crossing pages, pages released) and ends. Records are checked (count, tiling, a few
digests against hashlib) so that a profile of a broken run cannot pass for evidence.

  rocprofv3 --pmc SQ_INSTS_VALU ... -d OUT -- python scripts/r4_ring_pmc.py [GiB per stream] [streams]
"""
import hashlib
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["PBSGPU_RING_DEFER_SERVICE"] = "1"
os.environ.setdefault("PBSGPU_RING_BACKLOG_MIB", "0")

import numpy as np  # noqa: E402

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

GiB = 1 << 30


def main():
    per = float(sys.argv[1]) if len(sys.argv) > 1 else 48.0
    S = int(sys.argv[2]) if len(sys.argv) > 2 else 4
    nbytes = int(per * GiB) & ~15
    eng = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=1)
    ring = pbs_plus_amd.PageRing(eng, arena_bytes=int((per * S + 8) * GiB), max_streams=16)
    sids = [ring.open() for _ in range(S)]
    left = [nbytes] * S
    t0 = time.perf_counter()
    while any(left):
        for i, sid in enumerate(sids):
            if left[i]:
                want = min(left[i], 64 * ring.page_bytes)
                left[i] -= ring.fill(sid, 1000 + i, 4, want, final=(want == left[i]))
        ring.pump()
    while True:                                     # every round has run (nothing is hashed yet: the service is deferred)
        ring.pump()
        st = ring.stats()
        if st["rounds_done"] == st["rounds"] and st["bytes_enqueued"] == S * nbytes:
            break
        time.sleep(0.001)
    t_cut = time.perf_counter() - t0
    ring.quiesce()                                  # the service runs alone over the whole queue and ends
    st = ring.stats()
    total = 0
    ok = True
    for i, sid in enumerate(sids):
        parts = []
        while True:
            recs, fin = ring.poll(sid, 1 << 16)
            parts.append(recs.copy())
            if fin:
                break
        recs = np.concatenate(parts)
        total += recs.size
        ok &= bool(int(recs["end"][-1]) == nbytes and int(recs["size"].astype(np.int64).sum()) == nbytes)
        ring.close_stream(sid)
    print(json.dumps({"streams": S, "bytes_per_stream": nbytes, "records": int(total), "records_tile_their_streams": ok,
                      "cut_phase_s": round(t_cut, 3), "service_ms": st["service_ms_last"], "service_bytes": st["service_bytes_last"],
                      "service_GBps": round(st["service_bytes_last"] / max(st["service_ms_last"], 1e-9) / 1e6, 1),
                      "sha_cus": st["sha_cus"], "rounds": st["rounds"]}), flush=True)
    ring.close()
    eng.close()


if __name__ == "__main__":
    main()
