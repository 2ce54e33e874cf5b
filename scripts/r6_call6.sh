#!/bin/bash
# round 6, call 6: same-box A/B of the ROUND-5 library (_ref_r5: git 2c958e9) against the current one, the driver's command, alternating
out=gpurun_out/r6c6; mkdir -p $out
export PYTHONFAULTHANDLER=1
ROOT=$PWD
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[2], d['value'], d['ms_per_step'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, {k:v for k,v in r['single_file'].items() if k!='note'})
        if 'regime' in r: print('   regime', json.dumps({k:v for k,v in r['regime'].items() if k in ('feed_phase','drain','single_file')}))
PY
}
for i in 1 2; do
  ( cd $ROOT/_ref_r5 && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $ROOT/$out/r5_$i.json 2> $ROOT/$out/r5_$i.err ); line $out/r5_$i.json "round5 lib"
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/r6_$i.json 2> $out/r6_$i.err; line $out/r6_$i.json "round6 lib"
done
