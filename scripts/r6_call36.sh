#!/bin/bash
# round 6, call 36: how much backlog in front of the services does the line need? (default 128 MiB x 192 service CUs = 24 GiB:
# 30 ms of the drain)
out=gpurun_out/r6c36; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_RING_DEBUG=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
PY
  grep "occupancy\]" $out/b_$t.err | tail -1 | cut -c1-200
}
run base X=1
run bk16 PBSGPU_RING_BACKLOG_MIB=16384
run bk12 PBSGPU_RING_BACKLOG_MIB=12288
run bk8 PBSGPU_RING_BACKLOG_MIB=8192
run bk6 PBSGPU_RING_BACKLOG_MIB=6144
run base2 X=1
run bk12b PBSGPU_RING_BACKLOG_MIB=12288
run bk8b PBSGPU_RING_BACKLOG_MIB=8192
