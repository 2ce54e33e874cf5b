"""A/B of the candidate-scan kernels: parity against the oracle on small inputs, then the uncontended
kernel time on a large device-resident stream. Mode comes from PBSGPU_SCAN_MODE (read once per process).

  PBSGPU_SCAN_MODE=stream python scripts/ab_scan.py [GiB]
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import oracle as O  # noqa: E402  (checker only)
from pbs_plus_amd import Engine, buzhash  # noqa: E402
from tests.helpers import records_equal  # noqa: E402

mode = os.environ.get("PBSGPU_SCAN_MODE", "coop") + os.environ.get("PBSGPU_SCAN_DEPTH", "")
gib = float(sys.argv[1]) if len(sys.argv) > 1 else 16.0
for avg, n, kind in ((4096, 3_000_001, 3), (4096, 70_001, 0), (65536, 9_000_123, 1), (4 << 20, 80 << 20, 0),
                     (4 << 20, (200 << 20) + 77, 3)):
    eng = Engine(buzhash.NewConfig(avg))
    data = O.fill(n, 5, kind)
    cfg = O.new_config(avg)
    dbuf = eng.alloc(n + 128)
    dbuf.upload(data)
    got = eng.candidates(dbuf.ptr, n)
    want = O.candidates(cfg, data)
    assert np.array_equal(got, want), (mode, avg, n, kind, got[:8], want[:8], got.size, want.size)
    for lead in (1, 7, 64, 100):  # misaligned base
        assert np.array_equal(eng.candidates(dbuf.ptr + lead, n - lead), O.candidates(cfg, data[lead:])), (mode, avg, n, lead)
    dbuf.free()
    assert records_equal(eng.chunk_and_digest(data), O.chunk_and_digest(cfg, data)), (mode, avg, n)
    eng.close()
print(f"[{mode}] parity ok", flush=True)

eng = Engine(buzhash.NewConfig(4 << 20), inflight=1)
nbytes = int(gib * (1 << 30))
buf = eng.alloc(nbytes)
eng.fill(buf.ptr, nbytes, 2, 0)
best = 1e9
for it in range(4):
    tk = eng.submit(buf.ptr, nbytes=nbytes)
    eng.wait(tk)
    t = eng.timing(tk)
    eng.collect(tk)
    best = min(best, t["scan_ms"])
    print(f"[{mode}] {gib:g} GiB: scan {t['scan_ms']:.3f} ms = {nbytes / t['scan_ms'] / 1e9:.3f} TB/s, "
          f"resolve {t['resolve_ms']:.3f}, sha {t['sha_ms']:.1f}", flush=True)
print(f"[{mode}] best scan {best:.3f} ms = {nbytes / best / 1e9:.3f} TB/s")
