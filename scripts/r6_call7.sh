#!/bin/bash
# round 6, call 7: what in the round-6 service kernels costs the loaded chain 3.5 %? Same box: round-5 library, current, and
# variant builds (lib/variants: no probe; page walk of one register instead of eleven)
out=gpurun_out/r6c7; mkdir -p $out
export PYTHONFAULTHANDLER=1
ROOT=$PWD
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        rg = r.get('regime') or {}
        f = (rg.get('feed_phase') or {}).get('pair', {}).get('ns_per_block_step')
        print(sys.argv[2], d['value'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, r['single_file']['ms'], 'pair feed ns/step', f)
PY
}
run() { PBSGPU_LIB_PATH=$2 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$1.json 2> $out/$1.err; line $out/$1.json "$1"; }
V=$ROOT/pbs_plus_amd/lib/variants
for i in 1 2; do
  ( cd $ROOT/_ref_r5 && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $ROOT/$out/r5_$i.json 2> $ROOT/$out/r5_$i.err ); line $out/r5_$i.json "round5 lib     "
  run current_$i ""
  run noprobe_$i $V/libpbsgpu_noprobe.so
  run walk1_$i $V/libpbsgpu_walk1.so
  run noprobe_walk1_$i $V/libpbsgpu_noprobe_walk1.so
done
