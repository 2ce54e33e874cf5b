#!/bin/bash
# round 6, call 21: the DENSE pair service (eight waves per CU) in the ring: feed rate, drain, chains
out=gpurun_out/r6c21; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline') or {}
        reg=(r.get('regime') or {})
        print(sys.argv[2], d['value'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, 'one file', (r.get('single_file') or {}).get('ms'), (r.get('single_file') or {}).get('cut_ms'), 'feed', ((reg.get('feed_phase') or {}).get('pair') or {}).get('ns_per_block_step'), 'drain', ((reg.get('drain') or {}).get('pair') or {}).get('ns_per_block_step'), 'xp', ((reg.get('feed_phase') or {}).get('express') or {}).get('ns_per_block_step'), 'rounds', d['config'].get('rounds_in_timed_region'))
PY
}
timeout 300 python -m pytest tests/test_gpu_ring.py -m gpu -x -q > $out/pytest_ring_sparse.log 2>&1; tail -1 $out/pytest_ring_sparse.log
PBSGPU_RING_DENSE_SERVICE=1 timeout 600 python -m pytest tests/test_gpu_ring.py -m gpu -x -q > $out/pytest_ring_dense.log 2>&1; tail -1 $out/pytest_ring_dense.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_sparse.json 2> $out/bench_sparse.err; line $out/bench_sparse.json "sparse 176+16"
for cfg in "176 16" "168 24" "160 32" "176 8"; do
  set -- $cfg
  PBSGPU_RING_DENSE_SERVICE=1 PBSGPU_RING_XP_CUS=$2 timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $1 > $out/bench_dense_$1_$2.json 2> $out/bench_dense_$1_$2.err; line $out/bench_dense_$1_$2.json "dense $1+$2"
done
