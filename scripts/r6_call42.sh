#!/bin/bash
# round 6, call 42: the rate of each QUEUE's service in the steady state (PBSGPU_RING_TIER_TAG: the records say which queue their
# chunk went through; slopes of delivered bytes against time)
out=gpurun_out/r6c42; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_TRACE=1 PBSGPU_RING_TIER_TAG=1
run() { t=$1; st=$2; shift; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps $st --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$st steps $*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'], d['cpu_baseline'] if 'cpu_baseline' in d and d['cpu_baseline'] else '')
PY
  grep "delivered records" $out/b_$t.err | tail -1 | cut -c1-300
}
run base 40 X=1
run l64 40 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run l96 40 PBSGPU_RING_LANES_CUS=96 PBSGPU_RING_SHORT_BYTES=6291456
run d32 40 PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=32 PBSGPU_RING_SHORT_BYTES=3145728
run l32 40 PBSGPU_RING_LANES_CUS=32 PBSGPU_RING_SHORT_BYTES=6291456
