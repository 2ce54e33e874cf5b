# Round-2 profiles: the driver's exact bench command under rocprofv3 (kernel trace + stats), then separate PMC passes.
# Run on the GPU box through gpurun from the repo root; summaries are copied to profiles/ by hand afterwards.
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/gpurun_out/r2prof
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
set -x
rocprofv3 --kernel-trace --stats -d $OUT/kstats -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_traced.json 2> $OUT/bench_traced.err
tail -c 300 $OUT/bench_traced.err
python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 > $OUT/bench_plain.json 2> $OUT/bench_plain.err
rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_write.json 2> $OUT/pmc_write.err
rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_ANY SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_INSTS_LDS -d $OUT/pmc_sq -o s -- python3 $ROOT/bench.py --gpus 1 --steps 4 --warmup 0 --no-cpu-baseline > $OUT/pmc_sq.json 2> $OUT/pmc_sq.err
rocprofv3 --kernel-trace --stats -d $OUT/kstats_many -o many -- python3 $ROOT/bench.py --workload manyfiles --no-cpu-baseline > $OUT/many_traced.json 2> $OUT/many_traced.err
rocprofv3 --kernel-trace --stats -d $OUT/kstats_feed -o feed -- python3 $ROOT/bench.py --workload hostfeed --producers 8 --steps 12 > $OUT/feed_traced.json 2> $OUT/feed_traced.err
find $OUT -name "*.csv" | head -40
find $OUT -name "*.csv" -size +8M -delete
du -sh $OUT
