#!/bin/bash
# round 6, call 25: fewer cut CUs with the refill in stream order behind the previous scan (PBSGPU_RING_F_FILL_SERIAL)
out=gpurun_out/r6c25; mkdir -p $out
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
run() { # sha xp tag env...
  s=$1; x=$2; t=$3; shift 3
  env PBSGPU_RING_XP_CUS=$x "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $s > $out/b_${s}_${x}_$t.json 2> $out/b_${s}_${x}_$t.err; line $out/b_${s}_${x}_$t.json "sha$s+xp$x $*"
}
run 176 16 serial PBSGPU_RING_FILL_SERIAL=1
run 180 16 serial PBSGPU_RING_FILL_SERIAL=1
run 184 16 serial PBSGPU_RING_FILL_SERIAL=1
run 188 16 serial PBSGPU_RING_FILL_SERIAL=1
run 184 16 serial_r128 PBSGPU_RING_FILL_SERIAL=1 PBSGPU_RING_ROUND_PAGES=128
run 180 16 plain X=1
