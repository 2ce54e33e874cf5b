#!/bin/bash
# round 4, GPU call 11: scan of round n+1 beside the control kernel of round n (two cut streams, second scan set) — parity, then A/B
out=gpurun_out/r4c11; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_round4.py tests/test_gpu_xpair.py -m gpu -x -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|Error\|assert" $out/pytest.log | tail -5 | cut -c1-400
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'))" || tail -3 $out/$label.err
}
MiB=1048576
run overlap0 PBSGPU_RING_OVERLAP=0
run overlap1
run overlap1_sha196 PBSGPU_RING_SHA_CUS=196
run overlap1_sha200 PBSGPU_RING_SHA_CUS=200
run overlap1_xp16_long13 PBSGPU_RING_XP_CUS=16 PBSGPU_RING_LONG_BYTES=$((13*MiB))
run overlap1_xp16_long12 PBSGPU_RING_XP_CUS=16 PBSGPU_RING_LONG_BYTES=$((12*MiB))
run overlap0_again PBSGPU_RING_OVERLAP=0
