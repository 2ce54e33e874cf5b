#!/bin/bash
# round 6, call 49: the driver's line with every leg once more (the run of scripts/r6_final.sh fell into the large-round regime)
out=gpurun_out/r6final; mkdir -p $out
( time PBS_BENCH_HF_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r['valu']['frac'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file'], d['cpu_baseline'].get('records_match_gpu'), d['config']['rounds_in_timed_region'])
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle')} if isinstance(v,dict) else v)
PY
