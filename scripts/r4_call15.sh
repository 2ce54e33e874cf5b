#!/bin/bash
# round 4, GPU call 15: refill + scan streams at the lowest HIP priority (the control side of a round goes first) A/B
out=gpurun_out/r4c15; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -x -q --timeout 300 ) > $out/pytest.log 2>&1; grep -a "passed\|failed" $out/pytest.log | tail -3
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), d['config']['sha_service_cus'], d['config']['express_cus'], 'rounds', d['config']['rounds_in_timed_region'])" || tail -3 $out/$label.err
}
run prio0 PBSGPU_RING_CUT_PRIO=0
run prio1
run prio1_pair184_xp16 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run prio1_xp0_pair200 PBSGPU_RING_SHA_CUS=200 PBSGPU_RING_XP_CUS=0
run prio0_again PBSGPU_RING_CUT_PRIO=0
run prio1_again
