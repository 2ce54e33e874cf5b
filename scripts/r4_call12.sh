#!/bin/bash
# round 4, GPU call 12: whole suite + the driver's line with every leg on the new defaults (express service, scan/control overlap)
out=gpurun_out/r4c12; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED\|RCCL" $out/pytest.log | tail -8 | cut -c1-300
( time PBS_BENCH_HF_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['bound'], r['frac'], r['valu']['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file'], d.get('cpu_baseline',{}).get('records_match_gpu'), d['config'].get('express_cus'))
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','leg_seconds','write_phase')} if isinstance(v,dict) else v)
PY
tail -3 $out/bench_default.err | cut -c1-300
