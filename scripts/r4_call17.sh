#!/bin/bash
# round 4, GPU call 17: pair lanes help with long chunks sooner (spill at 1/8 of the express lanes), the stream writer's ring sends
# chunks >= half the maximum to its express service
out=gpurun_out/r4c17; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_xpair.py tests/test_gpu_round4.py -m gpu -x -q --timeout 300 ) > $out/pytest.log 2>&1; grep -a "passed\|failed" $out/pytest.log | tail -3
show() { python3 -c "
import json,sys
d=json.loads([l for l in open('$1') if l.startswith('{')][0]); r=d['roofline']
print('$2', d['value'], r.get('feed_phase'), r.get('frac_of_measured_h2d'), d.get('write_phase'), (d.get('cpu_baseline') or {}).get('records_match_gpu'))" ; }
timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/default.json 2>/dev/null; show $out/default.json default
timeout 300 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 > $out/ring_manyfiles.json 2>$out/ring_manyfiles.err; show $out/ring_manyfiles.json ring_manyfiles
PBSGPU_RING_XP_CUS=0 timeout 300 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/ring_manyfiles_xp0.json 2>/dev/null; show $out/ring_manyfiles_xp0.json ring_manyfiles_xp0
timeout 200 python bench.py --workload hostfeed --producers 1 --archives 1 --steps 96 --warmup 4 > $out/hf1.json 2>/dev/null; show $out/hf1.json hostfeed_1
timeout 200 python bench.py --workload hostfeed --producers 1 --archives 4 --steps 96 --warmup 4 > $out/hf1x4.json 2>/dev/null; show $out/hf1x4.json hostfeed_1x4
timeout 200 python bench.py --workload hostfeed --producers 8 --steps 32 --warmup 4 > $out/hf8.json 2>/dev/null; show $out/hf8.json hostfeed_8
