#!/bin/bash
# round 4, GPU call 19: run-to-run spread of the driver's command on the final code (two regimes: ~1300 small rounds / ~700 larger ones)
out=gpurun_out/r4c19; mkdir -p $out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), 'rounds', d['config']['rounds_in_timed_region'])" || tail -3 $out/$label.err
}
run final_1
run final_2
run final_3
run prio0_1 PBSGPU_RING_CUT_PRIO=0
run prio0_2 PBSGPU_RING_CUT_PRIO=0
run final_4
