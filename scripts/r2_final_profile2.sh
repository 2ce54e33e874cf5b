# kernel trace + stats of the final code: the driver's command, and the saturated regime (avg 64 KiB)
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2final2
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/k_default -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_traced.json 2> $OUT/bench_default_traced.err
$EXP stats $(db $OUT/k_default) $OUT/kernel_stats_bench_default.csv; $EXP trace $(db $OUT/k_default) $OUT/kernel_trace_bench_default.csv
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/k_avg64k -o bench -- python3 $ROOT/bench.py --gpus 1 --avg 65536 --steps 12 --warmup 2 --cpu-sample-gib 1 > $OUT/bench_avg64k_traced.json 2> $OUT/bench_avg64k_traced.err
$EXP stats $(db $OUT/k_avg64k) $OUT/kernel_stats_bench_avg64k.csv
timeout 300 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
for f in bench_default_traced bench_default bench_avg64k_traced; do python3 -c "
import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['serial_step_ms'], d.get('cpu_baseline',{}).get('records_match_gpu'))"; done
head -6 $OUT/kernel_stats_bench_default.csv | cut -c1-220; head -6 $OUT/kernel_stats_bench_avg64k.csv | cut -c1-220
