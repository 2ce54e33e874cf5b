#!/bin/bash
# round 4, GPU call 3: whole GPU suite on the one-engine code, the driver's line with every leg, service-CU sweep with the fused
# control kernel, profile evidence (kernel stats of the driver's command + PMC passes on the ring's own kernels)
out=gpurun_out/r4c3; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests -m gpu -x -q --timeout 400 ) > $out/pytest.log 2>&1; tail -30 $out/pytest.log | cut -c1-250
( time timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err; tail -3 $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file']['ms'], d.get('cpu_baseline',{}).get('records_match_gpu'))
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','leg_seconds','write_phase')} if isinstance(v,dict) else v)
PY
for cus in 196 200 204 208; do
  PBSGPU_RING_SHA_CUS=$cus timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/sweep_$cus.json 2>/dev/null
  python3 -c "
import json
for l in open('$out/sweep_$cus.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print('sha_cus $cus:', d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'])"
done
bash scripts/r4_profile.sh $out/prof 2>&1 | tail -40
