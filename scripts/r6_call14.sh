#!/bin/bash
# round 6, call 14: full rounds cost the cut side 1.5x more per byte than small ones (call 13's trace: 184 + 8 is stuck in
# full rounds at 675 GiB/s with the cut side 99.6 % busy). Does capping the round size remove the cliff? ONE box.
out=gpurun_out/r6c14; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[2], d['value'], {k:v for k,v in r['feed_phase'].items() if k!='note'}, 'one file', r['single_file']['ms'], r['single_file']['cut_ms'], 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
for rp in 256 128 64 32; do
for cfg in "0:" "184:8" "192:8"; do
  sha=${cfg%%:*}; xp=${cfg#*:}
  if [ -n "$xp" ]; then export PBSGPU_RING_XP_CUS=$xp; else unset PBSGPU_RING_XP_CUS; fi
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $sha --ring-round-pages $rp > $out/bench_${rp}_${sha}_${xp:-d}.json 2> $out/bench_${rp}_${sha}_${xp:-d}.err; line $out/bench_${rp}_${sha}_${xp:-d}.json "round_pages=$rp sha=$sha xp=${xp:-default}"
done
done
