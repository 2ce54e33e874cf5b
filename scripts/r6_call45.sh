#!/bin/bash
# round 6, call 45: where k_ring_control spends its time (RingRoundStatus::phase_ticks, summed by pbsgpu_ring_debug per round size)
out=gpurun_out/r6c45; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b.json 2> $out/b.err
grep -a "control kernel\|ring debug" $out/b.err | cut -c1-400
python3 - <<'PY'
import sys; sys.path.insert(0,'.')
import os
os.environ.pop('PBS_BENCH_RING_DEBUG',None)
PY
# the raw text of pbsgpu_ring_debug at the end of the run
python3 - $out/b.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file'])
PY
