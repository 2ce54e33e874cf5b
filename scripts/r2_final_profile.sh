# End-of-round evidence on the final code: parity suite, the driver's bench command under rocprofv3 (kernel trace +
# stats) and plain, and the saturated regime (avg 64 KiB, dense SHA form) under kernel trace and an SQ PMC pass.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2final
mkdir -p $OUT
cd $ROOT
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-600
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/k_default -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default_traced.json 2> $OUT/bench_default_traced.err
$EXP stats $(db $OUT/k_default) $OUT/kernel_stats_bench_default.csv; $EXP trace $(db $OUT/k_default) $OUT/kernel_trace_bench_default.csv
timeout 300 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/k_avg64k -o bench -- python3 $ROOT/bench.py --gpus 1 --avg 65536 --steps 12 --warmup 2 --cpu-sample-gib 1 > $OUT/bench_avg64k_traced.json 2> $OUT/bench_avg64k_traced.err
$EXP stats $(db $OUT/k_avg64k) $OUT/kernel_stats_bench_avg64k.csv
timeout 600 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_avg64k -o s -- python3 $ROOT/bench.py --gpus 1 --avg 65536 --steps 3 --warmup 0 --no-cpu-baseline > $OUT/pmc_avg64k.json 2> $OUT/pmc_avg64k.err
$EXP counters $(db $OUT/pmc_avg64k) $OUT/pmc_sq_bench_avg64k.csv
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
for f in bench_default_traced bench_default bench_avg64k_traced; do python3 -c "
import json,sys; d=json.loads(open('$OUT/$f.json').read().strip().splitlines()[-1]); print('$f', d['value'], d['ms_per_step'], d['roofline']['frac'], d['serial_step_ms'], d.get('cpu_baseline',{}).get('records_match_gpu'))"; done
head -8 $OUT/kernel_stats_bench_default.csv | cut -c1-200; head -8 $OUT/kernel_stats_bench_avg64k.csv | cut -c1-200; grep -i "sha256" $OUT/pmc_sq_bench_avg64k.csv | cut -c1-220
du -sh $OUT
