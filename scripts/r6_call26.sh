#!/bin/bash
# round 6, call 26: the LANES service (third tier: short chunks, one lane per chunk) — parity, then the driver's line
out=gpurun_out/r6c26; mkdir -p $out
export PYTHONFAULTHANDLER=1
PBSGPU_RING_LANES_CUS=2 timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -x -q > $out/pytest_ring_lanes.log 2>&1; tail -2 $out/pytest_ring_lanes.log
line() { python3 - "$1" "$2" <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', ((reg.get('feed_phase') or {}).get('pair') or {}).get('ns_per_block_step'), 'rounds', d['config'].get('rounds_in_timed_region'), 'sha_cus', d['config'].get('sha_service_cus'))
if not ok: print(sys.argv[2], 'no line'); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run() { # tag env...
  t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err; line $out/b_$t.json "$*"
}
run base X=1
run l32_6m PBSGPU_RING_LANES_CUS=32 PBSGPU_RING_SHORT_BYTES=6291456
run l64_6m PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run l96_6m PBSGPU_RING_LANES_CUS=96 PBSGPU_RING_SHORT_BYTES=6291456
run l64_4m PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=4194304
run l64_8m PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=8388608
