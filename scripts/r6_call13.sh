#!/bin/bash
# round 6, call 13: the unexplained CU split 184 + 8 (64 CUs left for the cut side, like the default 176 + 16, yet 12 % slower in
# round 5): the line, the regime probe and a kernel trace of both splits on ONE box; + the new GPU tests of this commit
out=gpurun_out/r6c13; mkdir -p $out
export PYTHONFAULTHANDLER=1
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
( time timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --timeout 200 -x -k "split or comm" ) > $out/pytest.log 2>&1; tail -3 $out/pytest.log | cut -c1-300
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; rg=r.get('regime') or {}
        g=lambda ph,s,k='ns_per_block_step': ((rg.get(ph) or {}).get(s) or {}).get(k)
        print(sys.argv[2], d['value'], {k:v for k,v in r['feed_phase'].items() if k!='note'}, 'one file', r['single_file']['ms'], r['single_file']['cut_ms'], 'rounds', d['config'].get('rounds_in_timed_region'), 'pair/express CUs', d['config'].get('sha_service_cus'), d['config'].get('express_cus'), 'pair ns feed/drain', g('feed_phase','pair'), g('drain','pair'), 'express', g('feed_phase','express'), g('drain','express'))
PY
}
for cfg in "0:" "184:8" "176:8" "192:8" "0:"; do
  sha=${cfg%%:*}; xp=${cfg#*:}
  if [ -n "$xp" ]; then export PBSGPU_RING_XP_CUS=$xp; else unset PBSGPU_RING_XP_CUS; fi
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus $sha > $out/bench_${sha}_${xp:-d}.json 2> $out/bench_${sha}_${xp:-d}.err; line $out/bench_${sha}_${xp:-d}.json "sha=$sha xp=${xp:-default}"
done
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
export PBSGPU_RING_XP_CUS=8
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/k_184_8 -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-sha-cus 184 > $OUT/bench_traced_184_8.json 2> $OUT/bench_traced_184_8.err
$EXP trace $(db $OUT/k_184_8) $OUT/kernel_trace_184_8.csv
python3 $ROOT/scripts/r5_trace_regimes.py $OUT/kernel_trace_184_8.csv
gzip -f $OUT/kernel_trace_184_8.csv
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
