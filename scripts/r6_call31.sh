#!/bin/bash
# round 6, call 31: whole GPU suite after the lanes service landed + default line
out=gpurun_out/r6c31; mkdir -p $out
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; grep -a "passed\|failed" $out/pytest_gpu.log | tail -2
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench.json 2> $out/bench.err; python3 - <<'PY'
import json
for l in open('gpurun_out/r6c31/bench.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['roofline']['frac'], d['cpu_baseline'].get('records_match_gpu'), d['cpu_baseline'].get('records_checked'))
PY
