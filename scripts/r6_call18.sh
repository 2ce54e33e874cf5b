#!/bin/bash
# round 6, call 18: the producers' block requests as GLOBAL loads with a real two-step prefetch (vmcnt(5)): suite, default
# line, A/B against the round-5 library on the same box
out=gpurun_out/r6c18; mkdir -p $out
export PYTHONFAULTHANDLER=1
timeout 1500 python -m pytest tests -m gpu -x -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], 'frac', r.get('frac'), {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', (reg.get('feed_phase') or {}).get('pair'), 'drain', (reg.get('drain') or {}).get('pair'), 'xp', (reg.get('feed_phase') or {}).get('express'))
PY
}
for i in 1 2 3; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err; line $out/bench_default_$i.json "new"
done
if [ -d _ref_r5 ]; then
  for i in 1 2; do
  (cd _ref_r5 && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > ../$out/bench_r5_$i.json 2> ../$out/bench_r5_$i.err); line $out/bench_r5_$i.json "round5"
  done
fi
for cu in 160 168 184; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $cu > $out/bench_sha$cu.json 2> $out/bench_sha$cu.err; line $out/bench_sha$cu.json "sha$cu"
done
timeout 900 python bench.py --gpus 1 --steps 20 --warmup 5 --no-cpu-baseline > $out/bench_extras.json 2> $out/bench_extras.err; python3 - <<'PY'
import json
for l in open('gpurun_out/r6c18/bench_extras.json'):
    if l.startswith('{'):
        d=json.loads(l); print('extras', d['value']); 
        for k,v in (d.get('legs') or d.get('extras') or {}).items():
            if isinstance(v,dict): print(' ',k, v.get('value') or v.get('GiBps'), v.get('records_match_gpu'))
PY
