#!/bin/bash
# round 5, call 7: the spill rule of the long-chunk queue on the DEFAULT line. With 16 express CUs the pair lanes take a long
# chunk whenever every express pair is busy (utilisation ~0.93 on random data: often) and such a chunk then runs a 0.44-0.47 s
# pair chain — the drain floor. An express pair comes free every ~0.35 ms (1 024 pairs / 0.36 s), so a long chunk that WAITS
# while fewer than L are queued loses <= L x 0.35 ms and gains 0.1 s. Sweep L (PBSGPU_RING_LONG_SPILL), a few service shapes,
# the poll period; configs[2] through the ring as the regression check (half its bytes are long chunks).
out=gpurun_out/r5c7; mkdir -p $out
export PYTHONFAULTHANDLER=1
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; f=r.get('feed_phase') or {}
        print('$2', d['value'], 'feed', f.get('GiBps'), 'drain', f.get('drain_seconds'), 'single', (r.get('single_file') or {}).get('ms'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
}
run() { # name, env...
  name=$1; shift
  env "$@" timeout 200 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$name.json 2> $out/$name.err; show $out/$name.json $name
}
run base A=1
run spill32 PBSGPU_RING_LONG_SPILL=32
run spill128 PBSGPU_RING_LONG_SPILL=128
run spill512 PBSGPU_RING_LONG_SPILL=512
run poll32 PBSGPU_RING_POLL_EVERY=32
run spill128_poll32 PBSGPU_RING_LONG_SPILL=128 PBSGPU_RING_POLL_EVERY=32
run xp24_long12_spill128 PBSGPU_RING_XP_CUS=24 PBSGPU_RING_LONG_BYTES=12582912 PBSGPU_RING_LONG_SPILL=128
run xp32_long10_spill256 PBSGPU_RING_XP_CUS=32 PBSGPU_RING_LONG_BYTES=10485760 PBSGPU_RING_LONG_SPILL=256
run base2 A=1
for sp in default 128; do
  e="A=1"; [ $sp != default ] && e="PBSGPU_RING_LONG_SPILL=$sp"
  env $e timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_$sp.json 2> $out/rmf_$sp.err; show $out/rmf_$sp.json "ring_manyfiles spill=$sp"
done
