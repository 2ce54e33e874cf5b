#!/bin/bash
# round 6, call 3: A/B of the page geometry on ONE box (61 tiles = rounds 3-5, 8 tiles = new default, 16 tiles), then a kernel
# trace of the new default: where does the cut side's time go with 8x more pages per round?
out=gpurun_out/r6c3; mkdir -p $out
export PYTHONFAULTHANDLER=1
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print(sys.argv[2], d['value'], d['ms_per_step'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, {k:v for k,v in r['single_file'].items() if k!='note'}, d.get('cpu_baseline',{}).get('records_match_gpu'), d['config'].get('arena_pages'), d['config'].get('page_bytes'), d['config'].get('rounds'))
PY
}
for t in 61 8 16 61 8; do
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-page-tiles $t > $out/bench_t$t.json 2> $out/bench_t$t.err
  line $out/bench_t$t.json tiles=$t
done
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
for t in 8 61; do
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/k_t$t -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline --ring-page-tiles $t > $OUT/bench_traced_t$t.json 2> $OUT/bench_traced_t$t.err
$EXP stats $(db $OUT/k_t$t) $OUT/kernel_stats_t$t.csv; $EXP trace $(db $OUT/k_t$t) $OUT/kernel_trace_t$t.csv
head -8 $OUT/kernel_stats_t$t.csv | cut -c1-200
python3 $ROOT/scripts/r5_trace_regimes.py $OUT/kernel_trace_t$t.csv
gzip -f $OUT/kernel_trace_t$t.csv
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
