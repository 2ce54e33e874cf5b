#!/bin/bash
# round 5, call 11: light-load express threshold (RingSource::long_lo: chunks from 10/16 of the maximum go express while fewer than
# 3/4 of the express pairs are taken) — ring / express / stream tests, then A/B on the driver's command (one file alone is the target)
out=gpurun_out/r5c11; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py tests/test_gpu_round4.py tests/test_gpu_round5.py tests/test_gpu_round3.py -x -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED\|Error" $out/pytest.log | tail -5 | cut -c1-300
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase') or {}
        print('$2', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', (r.get('single_file') or {}).get('ms'), 'cut', (r.get('single_file') or {}).get('cut_ms'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$name.json 2> $out/$name.err; show $out/$name.json $name
}
run lo10_a A=1
run off_a PBSGPU_RING_LONG_LO_BYTES=0
run lo10_b A=1
run off_b PBSGPU_RING_LONG_LO_BYTES=0
run lo8 PBSGPU_RING_LONG_LO_BYTES=8388608
run lo11 PBSGPU_RING_LONG_LO_BYTES=11534336
