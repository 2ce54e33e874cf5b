#!/bin/bash
# round 4, GPU call 13: RCCL beside the HIP runtime in use (torch imported AFTER libpbsgpu: the full suite's failing case); CU split
# re-sweep with the express service and the scan/control overlap in place; round size / rounds in flight
out=gpurun_out/r4c13; mkdir -p $out
export PYTHONFAULTHANDLER=1
timeout 200 python - > $out/comm_after_torch.log 2>&1 <<'PY'
import numpy as np
import pbs_plus_amd
from pbs_plus_amd import Comm, buzhash, RECORD_DTYPE
eng = pbs_plus_amd.Engine(buzhash.NewConfig(4096), device=0, inflight=1)
eng.chunk_and_digest(np.zeros(100000, dtype=np.uint8))
import torch, torch.distributed          # the order of the test process: libpbsgpu (system HIP) first, torch afterwards
torch.zeros(4, device="cuda").sum().item()
c = Comm(eng, Comm.unique_id(), 0, 1)
recs = np.zeros(1000, dtype=RECORD_DTYPE); recs["digest"][:, 0] = np.arange(1000) % 200; recs["size"] = 5
dup, st = c.dedup(recs, 4096)
print("comm after torch ok", int(dup.sum()), st)
print(sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip64' in l}))
PY
tail -3 $out/comm_after_torch.log | cut -c1-500
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), d['config']['sha_service_cus'], d['config']['express_cus'])" || tail -3 $out/$label.err
}
run default
run pair184_xp16 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run pair180_xp16 PBSGPU_RING_SHA_CUS=180 PBSGPU_RING_XP_CUS=16
run pair176_xp24_long12 PBSGPU_RING_SHA_CUS=176 PBSGPU_RING_XP_CUS=24 PBSGPU_RING_LONG_BYTES=12582912
run default_inflight4 PBSGPU_RING_MAX_INFLIGHT=4
run default_minround32 PBSGPU_RING_MIN_ROUND_PAGES=32
run default_minround128 PBSGPU_RING_MIN_ROUND_PAGES=128
