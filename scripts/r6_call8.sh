#!/bin/bash
# round 6, call 8: services back in the two-piece form + the one-wave probe: against the round-5 library on one box; and a deeper
# block FIFO in the ring's pair producer (D = 4 blocks in flight instead of 2: is it memory latency that slows the loaded producer?)
out=gpurun_out/r6c8; mkdir -p $out
export PYTHONFAULTHANDLER=1
ROOT=$PWD
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        rg = r.get('regime') or {}
        g = lambda ph, s: ((rg.get(ph) or {}).get(s) or {}).get('ns_per_block_step')
        print(sys.argv[2], d['value'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, r['single_file']['ms'], 'pair ns/step feed/drain/single', g('feed_phase','pair'), g('drain','pair'), g('single_file','pair'), 'express', g('feed_phase','express'), g('drain','express'), g('single_file','express'), 'sclk', ((rg.get('feed_phase') or {}).get('pair') or {}).get('sclk_mhz'))
PY
}
run() { PBSGPU_LIB_PATH=$2 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/$1.json 2> $out/$1.err; line $out/$1.json "$1"; }
V=$ROOT/pbs_plus_amd/lib/variants
for i in 1 2; do
  ( cd $ROOT/_ref_r5 && timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $ROOT/$out/r5_$i.json 2> $ROOT/$out/r5_$i.err ); line $out/r5_$i.json "round5 lib"
  run current_$i ""
  run d4_$i $V/libpbsgpu_d4.so
done
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py tests/test_gpu_dense.py -m gpu -q --timeout 300 -x ) > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-400
