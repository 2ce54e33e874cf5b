#!/bin/bash
# round 6, call 47: the resolve walk keeps 64 records in registers and stores them with one instruction: whole GPU suite, the control
# kernel's phase times, the line at the default split and with 4 / 8 CUs moved from the cut side to the pair service
out=gpurun_out/r6c47; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -m pytest tests -m gpu -q ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED" $out/pytest.log | tail -5 | cut -c1-300
export PBS_BENCH_RING_DEBUG=1
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  grep -a "control kernel" $out/b_$t.err | cut -c1-300
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'], r['single_file']['cut_ms'], 'rounds', d['config']['rounds_in_timed_region'], d['config']['sha_service_cus'], d['config']['express_cus'])
PY
}
run base X=1
run s180 PBSGPU_RING_SHA_CUS=180 PBSGPU_RING_XP_CUS=16
run s184 PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16
run base2 X=1
run s180b PBSGPU_RING_SHA_CUS=180 PBSGPU_RING_XP_CUS=16
run s178 PBSGPU_RING_SHA_CUS=178 PBSGPU_RING_XP_CUS=16
