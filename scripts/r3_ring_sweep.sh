#!/bin/bash
# round-3 sweep of the ring workload's knobs (run on the GPU box through gpurun); one JSON line per setting
out=${1:-gpurun_out/r3sweep}; shift
mkdir -p $out
run() { tag=$1; shift; timeout 300 python bench.py --steps 20 --warmup 5 --no-extras --no-cpu-baseline "$@" > $out/$tag.json 2> $out/$tag.err; python - $out/$tag.json $tag <<'PY'
import json,sys
try:
    d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0])
    r=d['roofline']
    print(sys.argv[2], 'value', d['value'], 'ms/step', d['ms_per_step'], 'svc_ms', r['service_launch_ms'], 'svc GB/s', r['achieved'], 'single ms', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'], 'cus', d['config']['sha_service_cus'])
except Exception as e:
    print(sys.argv[2], 'FAILED', e); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
"$@"
