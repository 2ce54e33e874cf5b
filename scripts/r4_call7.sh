#!/bin/bash
# round 4, GPU call 7: idle-lane poll period of the service's producers (1 = before), cut-ahead of several bulk streams, sha_cus re-sweep
out=gpurun_out/r4c7; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_round4.py -m gpu -x -q --timeout 300 ) > $out/pytest.log 2>&1; grep -a "passed\|failed" $out/pytest.log | tail -3
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>/dev/null
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'])"
}
run poll1_defer0 PBSGPU_RING_POLL_EVERY=1 PBSGPU_RING_LONE_DEFER_MS=0
run poll8_defer0 PBSGPU_RING_POLL_EVERY=8 PBSGPU_RING_LONE_DEFER_MS=0
run poll8_defer25 PBSGPU_RING_POLL_EVERY=8
run poll32_defer25 PBSGPU_RING_POLL_EVERY=32
run poll8_defer25_cus196 PBSGPU_RING_POLL_EVERY=8 PBSGPU_RING_SHA_CUS=196
run poll8_defer25_cus200 PBSGPU_RING_POLL_EVERY=8 PBSGPU_RING_SHA_CUS=200
run poll1_defer0_again PBSGPU_RING_POLL_EVERY=1 PBSGPU_RING_LONE_DEFER_MS=0
