#!/bin/bash
# round 5, call 14: soak — the whole -m gpu suite twice back to back on the final build, then the driver's command three times
out=gpurun_out/r5c14; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in 1 2; do
  ( time timeout 700 python -m pytest tests -m gpu -q --timeout 400 ) > $out/pytest_$i.log 2>&1; grep -a "passed\|failed\|FAILED\|Error" $out/pytest_$i.log | tail -4 | cut -c1-300
done
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase') or {}
        print('$2', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', (r.get('single_file') or {}).get('ms'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
for i in 1 2 3; do
  timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_$i.json 2> $out/bench_$i.err; show $out/bench_$i.json final_$i
done
