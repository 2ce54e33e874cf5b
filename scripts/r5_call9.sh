#!/bin/bash
# round 5, call 9: the round's tables staged into device memory by the round's first kernel (k_ring_stage) instead of being read
# from mapped host memory by every kernel of the round: whole -m gpu suite, then A/B on the driver's command
out=gpurun_out/r5c9; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED\|Error" $out/pytest.log | tail -5 | cut -c1-300
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase') or {}
        print('$2', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', (r.get('single_file') or {}).get('ms'), 'cut', (r.get('single_file') or {}).get('cut_ms'), 'rounds', d['config'].get('rounds_in_timed_region'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
}
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$name.json 2> $out/$name.err; show $out/$name.json $name
}
run stage1_a A=1
run stage0_a PBSGPU_RING_STAGE_INPUTS=0
run stage1_b A=1
run stage0_b PBSGPU_RING_STAGE_INPUTS=0
run stage1_pair184_xp16 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run stage1_pair184_xp8 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=8
run stage1_pair192_xp8 PBSGPU_RING_SHA_CUS=192 PBSGPU_RING_XP_CUS=8
timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf.json 2> $out/rmf.err; show $out/rmf.json ring_manyfiles
