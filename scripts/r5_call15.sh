#!/bin/bash
# round 5, call 15: contract checks on the final build — smoke(), and bench.py with NO flags (must default to one GPU and finish within minutes)
out=gpurun_out/r5c15; mkdir -p $out
( time timeout 200 python -c "import __graft_entry__ as g; g.smoke()" ) > $out/smoke.log 2>&1; tail -4 $out/smoke.log
( time timeout 400 python bench.py ) > $out/bench_noflags.json 2> $out/bench_noflags.err; tail -4 $out/bench_noflags.err
python3 - <<PY
import json
for l in open('$out/bench_noflags.json'):
    if l.startswith('{'):
        d=json.loads(l); print('no flags:', d['metric'][:40], d['value'], d['unit'], 'n_gpus', d['n_gpus'], 'steps', d['steps'], 'warmup', d['warmup'], 'ms_per_step', d['ms_per_step'], 'roofline', d['roofline']['bound'], d['roofline']['frac'], 'cpu', d['cpu_baseline']['value'], d['cpu_baseline']['kind'], 'legs', sorted((d.get('workloads') or {}).keys())[:4])
PY
