#!/bin/bash
# round 6, call 50: how often does the default line fall into the large-round regime? twelve fresh processes on one box
out=gpurun_out/r6c50; mkdir -p $out
for i in 1 2 3 4 5 6 7 8 9 10 11 12; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$i.json 2> $out/b_$i.err
  python3 - $out/b_$i.json $i <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
done
