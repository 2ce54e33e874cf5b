#!/bin/bash
# round 6, call 44: how long the first rounds on an idle ring are cut ahead of the services (lone_defer_ms, default 25): one file
# alone is cut at full width only while the services have not started
out=gpurun_out/r6c44; mkdir -p $out
export PYTHONFAULTHANDLER=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file'], 'rounds', d['config']['rounds_in_timed_region'])
PY
}
run d25 X=1
run d35 PBSGPU_RING_LONE_DEFER_MS=35
run d45 PBSGPU_RING_LONE_DEFER_MS=45
run d60 PBSGPU_RING_LONE_DEFER_MS=60
run d25b X=1
run d45b PBSGPU_RING_LONE_DEFER_MS=45
