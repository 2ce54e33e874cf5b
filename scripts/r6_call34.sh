#!/bin/bash
# round 6, call 34: does the lanes tier pay when the timed region is long (60 files: the drain is 6 % instead of 20 %)?
out=gpurun_out/r6c34; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 60 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
  grep "occupancy\]" $out/b_$t.err | tail -1 | cut -c1-300
}
run base X=1
run l64 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run base2 X=1
run l64b PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run l80 PBSGPU_RING_LANES_CUS=80 PBSGPU_RING_SHORT_BYTES=6291456
