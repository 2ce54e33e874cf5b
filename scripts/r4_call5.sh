#!/bin/bash
# round 4, GPU call 5: whole suite on the current code; the driver's line with every leg (host-feed legs traced); lone-stream deferral A/B
out=gpurun_out/r4c5; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 ) > $out/pytest.log 2>&1; tail -8 $out/pytest.log | cut -c1-250
( time PBS_BENCH_HF_TRACE=1 timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; grep "hostfeed trace" $out/bench_default.err | cut -c1-300
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file']['ms'], d.get('cpu_baseline',{}).get('records_match_gpu'))
        print('many_core', json.dumps(d['cpu_baseline'].get('many_core'))[:900])
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','leg_seconds','write_phase','results')} if isinstance(v,dict) and 'value' in v or isinstance(v,dict) and 'error' in v else v)
PY
for ms in 0 25; do
  PBSGPU_RING_LONE_DEFER_MS=$ms timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/lone_$ms.json 2>/dev/null
  python3 -c "
import json
d=json.loads([l for l in open('$out/lone_$ms.json') if l.startswith('{')][0]); r=d['roofline']; print('lone defer $ms ms:', d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'single file', r['single_file']['ms'])"
done
