#!/bin/bash
# round 6, call 23: the batch path's legs with the single-wave lanes form where the dense pair form is chosen today
out=gpurun_out/r6c23; mkdir -p $out
for w in corpus_dup rechunk manyfiles; do
  for mode in default lanes; do
    if [ $mode = lanes ]; then export PBSGPU_SHA_MODE=lanes; else unset PBSGPU_SHA_MODE; fi
    timeout 300 python bench.py --gpus 1 --workload $w --steps 4 --warmup 2 --no-cpu-baseline > $out/${w}_$mode.json 2> $out/${w}_$mode.err
    python3 - $out/${w}_$mode.json $w $mode <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); ok=True; print(sys.argv[2], sys.argv[3], d['value'], d.get('ms_per_step'), (d.get('serial_step_ms') or {}))
if not ok: print(sys.argv[2], sys.argv[3], 'no line')
PY
  done
done
