#!/bin/bash
# round 6, call 33: what refuses the feeder with the lanes tier on — the backlog gate or the arena? (free pages seen by the feeder)
out=gpurun_out/r6c33; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1 PBS_BENCH_RING_TRACE=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline $EXTRA > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$* $EXTRA" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
  grep "feeder\]\|occupancy\]\|ring trace\]" $out/b_$t.err | tail -4 | cut -c1-600
}
run base X=1
run l64 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run l64s4 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=4194304
run l64bk PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456 PBSGPU_RING_BACKLOG_MIB=64
run l64bk2 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456 PBSGPU_RING_BACKLOG_MIB=256
EXTRA="--arena-gib 200" run base_a200 X=1
