#!/bin/bash
# round 6, call 40: dense lanes tier + CUs moved to the cut side (the cut side co-limits at ~845 GiB/s of feed on 64 CUs)
out=gpurun_out/r6c40; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
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
D="PBSGPU_RING_DENSE_LANES=1 PBSGPU_RING_XP_CUS=16"
run base X=1
run A $D PBSGPU_RING_SHA_CUS=160 PBSGPU_RING_LANES_CUS=56 PBSGPU_RING_SHORT_BYTES=4194304
run B $D PBSGPU_RING_SHA_CUS=168 PBSGPU_RING_LANES_CUS=56 PBSGPU_RING_SHORT_BYTES=4194304
run C $D PBSGPU_RING_SHA_CUS=168 PBSGPU_RING_LANES_CUS=48 PBSGPU_RING_SHORT_BYTES=4194304
run Dd $D PBSGPU_RING_SHA_CUS=172 PBSGPU_RING_LANES_CUS=32 PBSGPU_RING_SHORT_BYTES=3145728
run E $D PBSGPU_RING_SHA_CUS=164 PBSGPU_RING_LANES_CUS=56 PBSGPU_RING_SHORT_BYTES=4194304
run F $D PBSGPU_RING_SHA_CUS=160 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=4194304
