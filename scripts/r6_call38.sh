#!/bin/bash
# round 6, call 38: the lanes tier with eight waves per CU (PBSGPU_RING_F_DENSE_LANES) for chunks of at most 3 MiB
out=gpurun_out/r6c38; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
( PBSGPU_RING_LANES_CUS=2 PBSGPU_RING_DENSE_LANES=1 timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -q -x --timeout 300 ) > $out/pytest_ring_dense_lanes.log 2>&1; tail -3 $out/pytest_ring_dense_lanes.log | cut -c1-300
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
  grep "occupancy\]" $out/b_$t.err | tail -1 | cut -c1-300
  grep "ring debug" $out/b_$t.err | tail -1 | cut -c1-200
}
run base X=1
run d40 PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=40 PBSGPU_RING_SHORT_BYTES=3145728
run d48 PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=48 PBSGPU_RING_SHORT_BYTES=3145728
run d32 PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=32 PBSGPU_RING_SHORT_BYTES=3145728
run d48s35 PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=48 PBSGPU_RING_SHORT_BYTES=3670016
run base2 X=1
run d40b PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_LANES_CUS=40 PBSGPU_RING_SHORT_BYTES=3145728
