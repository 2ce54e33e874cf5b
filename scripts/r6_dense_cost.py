"""What candidate-dense data costs (round 6: exact on-demand re-scan instead of PBSGPU_E_DENSITY). 2 GiB device-resident
streams at avg 4 MiB through the batch path (the same resolve walk as the ring's control kernel), pbsgpu_ticket_timing:
  random            ordinary bytes (parallel resolve)
  every position    a 64-byte period whose window hash passes the break test at every phase: every cut re-scans one row
  dense tile heads  3000 all-candidate bytes (more than a tile's 2176 slots) at the head of every 272 KiB scan tile, ordinary bytes behind: tiles overflow their
                    slots, the cut rule's search lands in the sparse rest and has to re-scan to the end of the tile (worst case)
Each result is compared with the oracle."""
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
import numpy as np  # noqa: E402

import dense_inputs as D  # noqa: E402
from oracle import oracle as O  # noqa: E402
from pbs_plus_amd import Engine, buzhash  # noqa: E402

O.build()
avg = 4 << 20
n = 2 << 30
cfg = O.new_config(avg)
allp = D.all_candidate_pattern(O.default_table())
TILE = 64 * 34 * 128
rnd = O.fill(n, 3, 0)
heads = rnd.copy()
for t in range(0, n - 3100, TILE):
    heads[t + 5:t + 3005] = np.tile(np.roll(allp, -((t // TILE * 7) % 64)), 48)[:3000]
cases = [("random", rnd), ("every position", np.tile(allp, n // 64)), ("dense tile heads", heads)]
eng = Engine(buzhash.NewConfig(avg), device=0, inflight=1)
buf = eng.alloc(n + 64)
for name, data in cases:
    buf.upload(data)
    want = O.chunk_and_digest(cfg, data)
    for rep in range(2):                      # (the second pass starts at the capacity the first one settled on)
        t0 = time.perf_counter()
        tk = eng.submit(buf, None, nbytes=n)
        tm = eng.timing(tk)
        got = eng.collect(tk)
        dt = time.perf_counter() - t0
        ok = got.size == want.size and np.array_equal(got["end"], want["end"]) and np.array_equal(got["digest"], want["digest"])
        print(f"{name:18s} pass {rep}: {got.size:6d} chunks, wall {dt * 1e3:8.1f} ms, scan {tm.get('scan_ms', 0):7.2f} ms, resolve "
              f"{tm.get('resolve_ms', 0):8.2f} ms, sha {tm.get('sha_ms', 0):8.1f} ms, re-runs {tm.get('retries')}, equals oracle: {ok}", flush=True)
buf.free()
eng.close()
