#!/bin/bash
# round 5, call 4: why does a host writer that starts its next archive while the previous one drains write at ~37 instead of ~47 GiB/s?
out=gpurun_out/r5c4; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_HF_TRACE=1
run() { name=$1; shift
  env "$@" timeout 200 python bench.py --workload hostfeed --producers 1 --steps 96 --warmup 4 --archives 4 --no-cpu-baseline > $out/$name.json 2> $out/$name.err
  python3 - <<PY
import json
for l in open('$out/$name.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d['value'], d['write_phase'])
PY
  grep -h "hostfeed trace" $out/$name.err | sed 's/.*archive of/  archive of/' | cut -c1-200
}
run base A=1
run nogate PBSGPU_RING_BACKLOG_MIB=0
run noxp PBSGPU_STREAM_XP_CUS=0
run sha64 PBSGPU_STREAM_SHA_CUS=64
run minround1 PBSGPU_RING_MIN_ROUND_PAGES=1
run copy8 PBSGPU_COPY_THREADS=8
run inflight1 PBSGPU_RING_MAX_INFLIGHT=1
