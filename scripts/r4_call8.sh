#!/bin/bash
# round 4, GPU call 8: the express SHA-256 form (two lanes per chunk) — parity as the batch path's only hash kernel and as the ring's
# express service; ring tests after the page-window change (256 pages per stream and round); bench A/B: express CUs x long threshold
out=gpurun_out/r4c8; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_xpair.py -m gpu -x -q --timeout 400 ) > $out/pytest_xpair.log 2>&1; grep -a "passed\|failed\|Error\|assert" $out/pytest_xpair.log | tail -8 | cut -c1-400
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_round4.py -m gpu -x -q --timeout 300 ) > $out/pytest_ring.log 2>&1; grep -a "passed\|failed" $out/pytest_ring.log | tail -3
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'))" || tail -3 $out/$label.err
}
run base
run xp16_long12 PBSGPU_RING_XP_CUS=16 PBSGPU_RING_LONG_BYTES=12582912
run xp24_long10 PBSGPU_RING_XP_CUS=24
run xp32_long10 PBSGPU_RING_XP_CUS=32
run xp32_long8 PBSGPU_RING_XP_CUS=32 PBSGPU_RING_LONG_BYTES=8388608
run xp24_long10_sha172 PBSGPU_RING_XP_CUS=24 PBSGPU_RING_SHA_CUS=172
