#!/bin/bash
# round 6, call 9: the whole -m gpu suite on the options-not-environment build + the driver's line with every leg
out=gpurun_out/r6c9; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1500 python -m pytest tests -m gpu -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED\|Error" $out/pytest.log | tail -12 | cut -c1-300
( time timeout 700 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], {k:v for k,v in r['feed_phase'].items() if k!='note'}, {k:v for k,v in r['single_file'].items() if k!='note'}, d.get('cpu_baseline',{}).get('records_match_gpu'))
        print('  regime', json.dumps({k:v for k,v in r['regime'].items() if k!='note'})[:900])
        print('  cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','kind','records_checked')})
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle')} if isinstance(v,dict) else v)
PY
tail -3 $out/bench_default.err | cut -c1-300
