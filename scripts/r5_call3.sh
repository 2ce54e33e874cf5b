#!/bin/bash
# round 5, call 3: can the synthetic refill run BESIDE the scan (lean producer + scan depth 2 leave 16 VGPRs per SIMD free)?
out=gpurun_out/r5c3; mkdir -p $out
export PYTHONFAULTHANDLER=1
( PBSGPU_SCAN_DEPTH=2 PBSGPU_RING_FILL_LEAN=1 timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py -q -x --timeout 400 -k "oracle or production or express" ) > $out/pytest_lean.log 2>&1
grep -a "passed\|failed\|FAILED\|Error" $out/pytest_lean.log | tail -5 | cut -c1-300
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$name.json 2> $out/$name.err
  python3 - <<PY
import json
for l in open('$out/$name.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase',{})
        print('$name', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', r['single_file'].get('ms'), r['single_file'].get('cut_ms'), 'cus', d['config'].get('sha_service_cus'), d['config'].get('express_cus'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
run base A=1
run depth2 PBSGPU_SCAN_DEPTH=2
run lean PBSGPU_RING_FILL_LEAN=1
run depth2_lean PBSGPU_SCAN_DEPTH=2 PBSGPU_RING_FILL_LEAN=1
run depth2_lean_184 PBSGPU_SCAN_DEPTH=2 PBSGPU_RING_FILL_LEAN=1 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run depth2_lean_192 PBSGPU_SCAN_DEPTH=2 PBSGPU_RING_FILL_LEAN=1 PBSGPU_RING_SHA_CUS=192 PBSGPU_RING_XP_CUS=16
run base2 A=1
for xp in 96 104 112; do
  PBSGPU_RING_XP_CUS=$xp timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_xp$xp.json 2> $out/rmf_xp$xp.err
  python3 - <<PY
import json
for l in open('$out/rmf_xp$xp.json'):
    if l.startswith('{'):
        d=json.loads(l); print('ring_manyfiles xp=$xp', d['value'], d['roofline'].get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
done
timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_auto.json 2> $out/rmf_auto.err
python3 - <<PY
import json
for l in open('$out/rmf_auto.json'):
    if l.startswith('{'):
        d=json.loads(l); print('ring_manyfiles auto', d['value'], d['roofline'].get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
