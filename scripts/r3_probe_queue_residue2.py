#!/usr/bin/env python3
"""Follow-up of r3_probe_queue_residue.py: the legs that lose 20-30 % behind a host-fed leg all keep SEVERAL kernels in
flight. Runs P payload streams that write MiB each (then closes everything) and then bench.py's batch workload (4 slots
in flight) in the same process.  usage: r3_probe_queue_residue2.py P MiB_each [bench args...]"""
import os
import runpy
import sys
import threading

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np  # noqa: E402
import torch  # noqa: E402,F401  (bench.py's import order: torch first)

import pbs_plus_amd  # noqa: E402
from pbs_plus_amd import buzhash  # noqa: E402

P, mib = int(sys.argv[1]), int(sys.argv[2])
if P:
    eng = pbs_plus_amd.Engine(buzhash.NewConfig(4 << 20), device=0, inflight=2)
    src = np.random.default_rng(3).integers(0, 256, 32 << 20, dtype=np.uint8)

    def one(i):
        st = pbs_plus_amd.PayloadStream(eng, 256 << 20)
        for _ in range(max(1, mib // 32)):
            st.write(src)
        st.finish()
        st.poll()
        st.close()
    ths = [threading.Thread(target=one, args=(i,)) for i in range(P)]
    for t in ths:
        t.start()
    for t in ths:
        t.join()
    eng.close()
sys.argv = [os.path.join(ROOT, "bench.py")] + (sys.argv[3:] or ["--workload", "stream64g", "--steps", "8", "--warmup", "4", "--no-extras",
                                                                "--no-cpu-baseline"])
runpy.run_path(sys.argv[0], run_name="__main__")
