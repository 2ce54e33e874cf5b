#!/bin/bash
# round 4, GPU call 16: CU split around the default with the low-priority bulk streams
out=gpurun_out/r4c16; mkdir -p $out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), d['config']['sha_service_cus'], d['config']['express_cus'], 'rounds', d['config']['rounds_in_timed_region'])" || tail -3 $out/$label.err
}
run default
run pair168_xp16 PBSGPU_RING_SHA_CUS=168 PBSGPU_RING_XP_CUS=16
run pair192_xp0 PBSGPU_RING_XP_CUS=0
run pair184_xp8_long14 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=8 PBSGPU_RING_LONG_BYTES=14680064
run pair176_xp16_minround32 PBSGPU_RING_MIN_ROUND_PAGES=32
