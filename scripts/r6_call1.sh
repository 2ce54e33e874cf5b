#!/bin/bash
# round 6, call 1: the exact dense path (tests/test_gpu_dense.py + the rewritten ring / stream tests), then the driver's line
out=gpurun_out/r6c1; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_dense.py tests/test_gpu_ring.py tests/test_gpu_round4.py -m gpu -q --timeout 300 -x ) > $out/pytest_dense.log 2>&1; tail -25 $out/pytest_dense.log | cut -c1-400
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r.get('feed_phase'), r['single_file'], d.get('cpu_baseline',{}).get('records_match_gpu'))
PY
tail -3 $out/bench_default.err | cut -c1-300
