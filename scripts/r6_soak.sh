#!/bin/bash
# round 6, soak on the final build: the whole -m gpu suite three times back to back, the driver's command five times
out=gpurun_out/r6soak2; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in 1 2 3; do
  ( time timeout 1200 python -m pytest tests -m gpu -q ) > $out/pytest_$i.log 2>&1; echo "suite $i: $(grep -a 'passed\|failed' $out/pytest_$i.log | tail -1 | cut -c1-120)"
done
for i in 1 2 3 4 5; do
  timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_$i.json 2> $out/bench_$i.err
  python3 - $out/bench_$i.json $i <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; g=r['regime']
        print('bench', sys.argv[2], d['value'], 'feed', r['feed_phase']['GiBps'], 'drain', r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'], 'pair ns', g['feed_phase']['pair']['ns_per_block_step'], 'sclk', g['feed_phase']['pair']['sclk_mhz'], 'match', d['cpu_baseline'].get('records_match_gpu'))
PY
done
