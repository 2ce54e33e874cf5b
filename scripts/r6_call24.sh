#!/bin/bash
# round 6, call 24: fewer cut CUs with SMALL rounds (cache-resident refill -> scan) — can 56 cut CUs keep up with 200 service CUs?
out=gpurun_out/r6c24; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', ((reg.get('feed_phase') or {}).get('pair') or {}).get('ns_per_block_step'), 'rounds', d['config'].get('rounds_in_timed_region'))
if not ok: print(sys.argv[2], 'no line')
PY
}
run() { # sha xp round_pages [extra env]
  env PBSGPU_RING_XP_CUS=$2 $4 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $1 --ring-round-pages $3 > $out/b_$1_$2_$3_$5.json 2> $out/b_$1_$2_$3_$5.err; line $out/b_$1_$2_$3_$5.json "sha$1+xp$2 round_pages=$3 $4"
}
run 176 16 0 "" a
run 176 16 64 "" a
run 184 16 64 "" a
run 184 16 32 "" a
run 184 16 128 "" a
run 184 16 64 "PBSGPU_RING_MIN_ROUND_PAGES=16" b
run 184 16 64 "PBSGPU_RING_MAX_INFLIGHT=4" c
run 192 16 64 "" a
