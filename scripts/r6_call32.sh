#!/bin/bash
# round 6, call 32: main-queue positions by ticket (no void positions): parity of the ring tests with and without the lanes tier,
# then the line with 0 / 64 / 96 lanes CUs
out=gpurun_out/r6c32; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
( timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_dense.py -m gpu -q -x --timeout 300 ) > $out/pytest_ring.log 2>&1; tail -3 $out/pytest_ring.log | cut -c1-300
( PBSGPU_RING_LANES_CUS=2 timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -q -x --timeout 300 ) > $out/pytest_ring_lanes.log 2>&1; tail -3 $out/pytest_ring_lanes.log | cut -c1-300
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
  grep "feeder\]\|occupancy\]" $out/b_$t.err | tail -2 | cut -c1-400
}
run base X=1
run l64 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run l96 PBSGPU_RING_LANES_CUS=96 PBSGPU_RING_SHORT_BYTES=6291456
run base2 X=1
run l64b PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
