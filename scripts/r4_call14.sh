#!/bin/bash
# round 4, GPU call 14: the synthetic producer no longer re-reads its page entry from host memory inside the generator loop;
# CU split re-sweep behind it; RCCL beside the HIP runtime in use with torch imported after libpbsgpu
out=gpurun_out/r4c14; mkdir -p $out
export PYTHONFAULTHANDLER=1
timeout 200 python - > $out/comm_after_torch.log 2>&1 <<'PY'
import numpy as np
import pbs_plus_amd
from pbs_plus_amd import Comm, buzhash, RECORD_DTYPE
eng = pbs_plus_amd.Engine(buzhash.NewConfig(4096), device=0, inflight=1)
eng.chunk_and_digest(np.zeros(100000, dtype=np.uint8))
import torch, torch.distributed          # the order of the test process: libpbsgpu (system HIP) first, torch (CPU use only) afterwards
c = Comm(eng, Comm.unique_id(), 0, 1)
recs = np.zeros(1000, dtype=RECORD_DTYPE); recs["digest"][:, 0] = np.arange(1000) % 200; recs["size"] = 5
dup, st = c.dedup(recs, 4096)
print("comm after torch ok", int(dup.sum()), st)
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip64' in l}))
PY
tail -2 $out/comm_after_torch.log | cut -c1-500
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py -m gpu -x -q --timeout 300 ) > $out/pytest.log 2>&1; grep -a "passed\|failed" $out/pytest.log | tail -3
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), d['config']['sha_service_cus'], d['config']['express_cus'])" || tail -3 $out/$label.err
}
run default
run pair184_xp16 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run pair192_xp16 PBSGPU_RING_SHA_CUS=192 PBSGPU_RING_XP_CUS=16
run pair200_xp16 PBSGPU_RING_SHA_CUS=200 PBSGPU_RING_XP_CUS=16
run pair200_xp0 PBSGPU_RING_SHA_CUS=200 PBSGPU_RING_XP_CUS=0
run pair208_xp0 PBSGPU_RING_SHA_CUS=208 PBSGPU_RING_XP_CUS=0
