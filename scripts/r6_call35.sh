#!/bin/bash
# round 6, call 35: probe counters through the round status block (no hipMemcpy beside the services): whole GPU suite + the line twice
out=gpurun_out/r6c35; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED" $out/pytest.log | tail -5 | cut -c1-300
run() { t=$1; shift
  env "$@" timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/b_$t.json 2> $out/b_$t.err
  python3 - $out/b_$t.json "$*" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; print(sys.argv[2], d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], 'one file', r['single_file']['ms'], 'rounds', d['config']['rounds_in_timed_region'])
        g=r['regime']; print({ph:{k:(v['ns_per_block_step'],v['sclk_mhz'],v['wave_steps_sampled']) for k,v in g[ph].items()} for ph in ('feed_phase','drain','single_file')})
PY
}
run base X=1
run base2 X=1
