#!/bin/bash
# round 5, call 8: which of the two regimes of the default line (≈1 350 small rounds / ≈700-800 larger ones, -1 %) a run lands in —
# four runs each with the synthetic refill on its own stream (default) and in stream order behind the previous scan
out=gpurun_out/r5c8; mkdir -p $out
export PYTHONFAULTHANDLER=1
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase') or {}
        print('$2', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', (r.get('single_file') or {}).get('ms'), 'cut', (r.get('single_file') or {}).get('cut_ms'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$name.json 2> $out/$name.err; show $out/$name.json $name
}
for i in 1 2 3 4; do
  run default_$i A=1
  run fill_serial_$i PBSGPU_RING_FILL_SERIAL=1
done
