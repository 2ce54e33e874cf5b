#!/bin/bash
# round 6, call 20: with the services at 1.74 us per block step — where does the CU split want to be now?
out=gpurun_out/r6c20; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', ((reg.get('feed_phase') or {}).get('pair') or {}).get('ns_per_block_step'), 'xp', ((reg.get('feed_phase') or {}).get('express') or {}).get('ns_per_block_step'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
for cfg in "176 16" "184 16" "192 16" "200 16" "192 8" "184 12"; do
  set -- $cfg
  PBSGPU_RING_XP_CUS=$2 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $1 > $out/bench_$1_$2.json 2> $out/bench_$1_$2.err; line $out/bench_$1_$2.json "sha$1+xp$2"
done
