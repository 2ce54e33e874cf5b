#!/bin/bash
# round 6, call 27: the lanes service with CUs handed back to the cut side
out=gpurun_out/r6c27; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
ok=False
for l in open(sys.argv[1]):
    if l.startswith('{'):
        ok=True
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', ((reg.get('feed_phase') or {}).get('pair') or {}).get('ns_per_block_step'), 'rounds', d['config'].get('rounds_in_timed_region'), 'pair_cus', d['config'].get('sha_service_cus'))
if not ok: print(sys.argv[2], 'no line'); print(open(sys.argv[1].replace('.json','.err')).read()[-600:])
PY
}
run() { # tag sha env...
  t=$1; s=$2; shift 2
  env PBSGPU_RING_XP_CUS=16 "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $s > $out/b_$t.json 2> $out/b_$t.err; line $out/b_$t.json "sha_cus=$s $*"
}
run a 168 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run b 168 PBSGPU_RING_LANES_CUS=96 PBSGPU_RING_SHORT_BYTES=6291456
run c 160 PBSGPU_RING_LANES_CUS=64 PBSGPU_RING_SHORT_BYTES=6291456
run d 160 PBSGPU_RING_LANES_CUS=96 PBSGPU_RING_SHORT_BYTES=6291456
run e 172 PBSGPU_RING_LANES_CUS=80 PBSGPU_RING_SHORT_BYTES=6291456
run f 168 X=1
