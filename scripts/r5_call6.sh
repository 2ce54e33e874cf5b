#!/bin/bash
# round 5, call 6: with a large express service, let long chunks WAIT for an express pair (queue-length rule) instead of spilling to
# pair lanes as soon as every express pair is busy; + the driver's full line with the host-fed legs in their own processes
out=gpurun_out/r5c6; mkdir -p $out
export PYTHONFAULTHANDLER=1
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['value'], d['roofline'].get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
}
for sp in 0 104 416 1664; do
  PBSGPU_RING_LONG_SPILL=$sp timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_sp$sp.json 2> $out/rmf_sp$sp.err; show $out/rmf_sp$sp.json "ring_manyfiles spill=$sp"
done
timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 > $out/rmf_default.json 2> $out/rmf_default.err; show $out/rmf_default.json "ring_manyfiles default(+oracle)"
grep -o '"records_match_gpu": [a-z]*' $out/rmf_default.json
( time PBS_BENCH_HF_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file'].get('ms'), d.get('cpu_baseline',{}).get('records_match_gpu'))
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','service_cus','leg_seconds')} if isinstance(v,dict) else v)
PY
tail -3 $out/bench_default.err
