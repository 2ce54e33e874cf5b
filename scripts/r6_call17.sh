#!/bin/bash
# round 6, call 17: round cap only with two or more streams waiting: default line + one file alone; then call 16's counters
out=gpurun_out/r6c17; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
for i in 1 2 3; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err; line $out/bench_default_$i.json "default"
done
PBSGPU_RING_XP_CUS=8 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus 184 > $out/bench_184_8.json 2> $out/bench_184_8.err; line $out/bench_184_8.json "184+8"
bash scripts/r6_call16.sh
