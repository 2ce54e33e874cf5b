#!/bin/bash
# round 6, call 2: small pages + progressive release (multi-page descriptors) in the pair and express services: parity, then the line
out=gpurun_out/r6c2; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py tests/test_gpu_dense.py -m gpu -q --timeout 300 -x ) > $out/pytest.log 2>&1; tail -25 $out/pytest.log | cut -c1-600
for i in 1 2; do
( time timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras ) > $out/bench_default_$i.json 2> $out/bench_default_$i.err
python3 - <<PY
import json
for l in open('$out/bench_default_$i.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, {k:v for k,v in r['single_file'].items() if k!='note'}, d.get('cpu_baseline',{}).get('records_match_gpu'), d['config'].get('arena_pages'), d['config'].get('page_bytes'))
PY
tail -3 $out/bench_default_$i.err | cut -c1-300
done
