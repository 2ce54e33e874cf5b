#!/bin/bash
# round 4, GPU call 9: idle pairs no longer nap beside a working sibling (shared block barrier); express service: CUs x threshold
out=gpurun_out/r4c9; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_xpair.py tests/test_gpu_ring.py -m gpu -x -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|Error\|assert" $out/pytest.log | tail -5 | cut -c1-400
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'))" || tail -3 $out/$label.err
}
MiB=1048576
run base
run xp14_long13 PBSGPU_RING_XP_CUS=14 PBSGPU_RING_LONG_BYTES=$((13*MiB))
run xp18_long12 PBSGPU_RING_XP_CUS=18 PBSGPU_RING_LONG_BYTES=$((12*MiB))
run xp24_long11 PBSGPU_RING_XP_CUS=24 PBSGPU_RING_LONG_BYTES=$((11*MiB))
run xp32_long10 PBSGPU_RING_XP_CUS=32
run base_again
