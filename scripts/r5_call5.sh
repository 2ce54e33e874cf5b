#!/bin/bash
# round 5, call 5: which earlier leg of the driver's line slows the 4-archive host-fed leg (34.7 in the line, 42.2 alone)?
out=gpurun_out/r5c5; mkdir -p $out
export PYTHONFAULTHANDLER=1 PBS_BENCH_HF_TRACE=1
run() { name=$1; shift
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline "$@" > $out/$name.json 2> $out/$name.err
  python3 - <<PY
import json
for l in open('$out/$name.json'):
    if l.startswith('{'):
        d=json.loads(l); print('$name', d['value'])
        for k,v in (d.get('workloads') or {}).items():
            if isinstance(v,dict) and ('value' in v or 'error' in v): print('   ', k, v.get('value'), v.get('error'), v.get('write_phase'))
PY
  grep -h "hostfeed trace" $out/$name.err | sed 's/.*archive of/      archive of/' | cut -c1-120
}
run only_hf1 --extras hostfeed1
run ringlegs_hf1 --extras ring_manyfiles,ring_corpus_dup,ring_rechunk,hostfeed1
run batchlegs_hf1 --extras batch,manyfiles,corpus_dup,rechunk,hostfeed1
