#!/bin/bash
# Round-3 evidence on the final code (run on the GPU box through gpurun):
#   1. the whole -m gpu parity suite
#   2. the driver's bench command (ring workload) under rocprofv3 --kernel-trace --stats, and plain
#   3. FETCH_SIZE / WRITE_SIZE PMC passes (separate runs) on the batch path of the same kernels — the profiler serialises
#      dispatches while it collects counters, so a run with the persistent ring service cannot be counted (the cut rounds
#      would wait for a kernel that only ends on request); k_scan3 and the SHA producer read the same bytes either way
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/${1:-gpurun_out/r3final}
mkdir -p $OUT
cd $ROOT
(time timeout 1200 python -X faulthandler -m pytest tests -x -q -m gpu) > $OUT/pytest.log 2>&1; tail -4 $OUT/pytest.log | cut -c1-300
(timeout 120 python -c "import __graft_entry__ as g; g.smoke(); print('smoke ok')") 2>&1 | tail -2
(timeout 400 python scripts/r3_ring_stress.py ${STRESS_ITERS:-300}) > $OUT/ring_stress.log 2>&1; tail -2 $OUT/ring_stress.log | cut -c1-300
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
timeout 600 rocprofv3 --kernel-trace --stats -d $OUT/k_default -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/bench_default_traced.json 2> $OUT/bench_default_traced.err
$EXP stats $(db $OUT/k_default) $OUT/kernel_stats_bench_default.csv; $EXP trace $(db $OUT/k_default) $OUT/kernel_trace_bench_default.csv
(time timeout 400 python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5) > $OUT/bench_default.json 2> $OUT/bench_default.err
timeout 600 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 $ROOT/bench.py --gpus 1 --workload stream64g --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_fetch.json 2> $OUT/pmc_fetch.err
$EXP counters $(db $OUT/pmc_fetch) $OUT/pmc_fetch_size_batch_path.csv
timeout 600 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 $ROOT/bench.py --gpus 1 --workload stream64g --steps 3 --warmup 0 --no-cpu-baseline --no-extras > $OUT/pmc_write.json 2> $OUT/pmc_write.err
$EXP counters $(db $OUT/pmc_write) $OUT/pmc_write_size_batch_path.csv
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
gzip -f $OUT/kernel_trace_bench_default.csv
for f in bench_default_traced bench_default; do python3 -c "
import json; d=json.loads([l for l in open('$OUT/$f.json') if l.startswith('{')][0]); r=d['roofline']; print('$f', d['value'], d['ms_per_step'], r['frac'], r['service_launch_ms'], r.get('feed_phase'), d.get('cpu_baseline',{}).get('records_match_gpu'))
w=d.get('workloads') or {}
for k,v in w.items(): print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','leg_seconds')} if isinstance(v,dict) else v)"; done
head -6 $OUT/kernel_stats_bench_default.csv | cut -c1-180
grep -i "scan3\|sha256_pair\|k_fill" $OUT/pmc_fetch_size_batch_path.csv | cut -c1-200
grep -i "scan3\|sha256_pair\|k_fill" $OUT/pmc_write_size_batch_path.csv | cut -c1-200
du -sh $OUT
