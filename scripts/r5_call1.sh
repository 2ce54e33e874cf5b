#!/bin/bash
# round 5, call 1: the round's new tests + the ring / stream tests they touch, the driver's line on this box (baseline of the
# round), and configs[2] through the ring with larger express shares (half its bytes sit in 16 MiB chunks)
out=gpurun_out/r5c1; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 900 python -m pytest tests/test_gpu_round5.py tests/test_gpu_ring.py tests/test_gpu_round4.py tests/test_gpu_xpair.py -q -x --timeout 400 -s ) > $out/pytest.log 2>&1
grep -a "passed\|failed\|FAILED\|Error\|1.06 TiB" $out/pytest.log | tail -12 | cut -c1-300
( time timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file'], d.get('cpu_baseline',{}).get('records_match_gpu'))
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','feed_phase','write_phase')} if isinstance(v,dict) else v)
PY
for xp in 16 48 80 112; do
  PBSGPU_RING_XP_CUS=$xp timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_xp$xp.json 2> $out/rmf_xp$xp.err
  python3 - <<PY
import json
for l in open('$out/rmf_xp$xp.json'):
    if l.startswith('{'):
        d=json.loads(l); print('ring_manyfiles xp=$xp', d['value'], d['roofline'].get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
done
