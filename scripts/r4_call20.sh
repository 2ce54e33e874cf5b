#!/bin/bash
# round 4, GPU call 20: does a smaller cap on the pages per round keep the ring in the "small rounds" regime?
out=gpurun_out/r4c20; mkdir -p $out
run() { # label, env...
  local label=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$label.json 2>$out/$label.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$label.json') if l.startswith('{')][0]); r=d['roofline']; print('$label:', d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'], 'cut', r['single_file'].get('cut_ms'), 'rounds', d['config']['rounds_in_timed_region'])" || tail -3 $out/$label.err
}
run rp64_1 PBSGPU_RING_ROUND_PAGES=64
run default_1
run rp64_2 PBSGPU_RING_ROUND_PAGES=64
run rp96_1 PBSGPU_RING_ROUND_PAGES=96
