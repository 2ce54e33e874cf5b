#!/bin/bash
# round 6, call 48: the large-round attractor (one default run of the final evidence fell into it: 617 rounds, feed 771, 630 GiB/s) —
# rounds in flight and round cap, at the default split and at 180 + 16 (which always runs at the cap)
out=gpurun_out/r6c48; mkdir -p $out
export PYTHONFAULTHANDLER=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'], r['single_file']['cut_ms'], 'rounds', d['config']['rounds_in_timed_region'], d['config']['sha_service_cus'], d['config']['express_cus'])
PY
}
S="PBSGPU_RING_SHA_CUS=180 PBSGPU_RING_XP_CUS=16"
run base X=1
run mi1 PBSGPU_RING_MAX_INFLIGHT=1
run mi2 PBSGPU_RING_MAX_INFLIGHT=2
run rp128 PBSGPU_RING_ROUND_PAGES=128
run s180_mi1 $S PBSGPU_RING_MAX_INFLIGHT=1
run s180_mi2 $S PBSGPU_RING_MAX_INFLIGHT=2
run s180_rp128 $S PBSGPU_RING_ROUND_PAGES=128
run s180_rp128_mi1 $S PBSGPU_RING_ROUND_PAGES=128 PBSGPU_RING_MAX_INFLIGHT=1
run s184_rp128_mi1 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16 PBSGPU_RING_ROUND_PAGES=128 PBSGPU_RING_MAX_INFLIGHT=1
