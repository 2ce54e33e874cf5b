#!/bin/bash
# round 6, call 46: wave-wide search for the walk's first candidate: whole GPU suite, then the control kernel's phase times and the line
out=gpurun_out/r6c46; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED" $out/pytest.log | tail -5 | cut -c1-300
export PBS_BENCH_RING_DEBUG=1
for i in 1 2; do
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b$i.json 2> $out/b$i.err
grep -a "control kernel" $out/b$i.err | cut -c1-400
python3 - $out/b$i.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file'], d['config']['rounds_in_timed_region'])
PY
done
