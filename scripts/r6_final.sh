#!/bin/bash
# round 6, final evidence on the final code: whole -m gpu suite, the driver's line with every leg, rocprofv3 kernel stats of the
# driver's command, SQ + FETCH_SIZE passes on the ring's own kernels (both services as ordinary dispatches: scripts/r4_ring_pmc.py)
out=gpurun_out/r6final; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1200 python -m pytest tests -m gpu -q --timeout 400 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED" $out/pytest.log | tail -5 | cut -c1-300
( time PBS_BENCH_HF_TRACE=1 timeout 600 python bench.py --gpus 1 --steps 20 --warmup 5 ) > $out/bench_default.json 2> $out/bench_default.err
python3 - <<PY
import json
for l in open('$out/bench_default.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], r['bound'], r['frac'], r['valu']['frac'], r.get('feed_phase',{}).get('GiBps'), r.get('feed_phase',{}).get('drain_seconds'), r['single_file'], d.get('cpu_baseline',{}).get('records_match_gpu'), d['config'].get('express_cus'))
        print('  cpu', {k:v for k,v in d['cpu_baseline'].items() if k in ('value','cores','kind','records_checked')}, json.dumps(d['cpu_baseline'].get('many_core'))[:300])
        for k,v in (d.get('workloads') or {}).items():
            print('  ', k, {kk:v[kk] for kk in v if kk in ('value','error','records_match_gpu','frac_of_measured_h2d','records_match_oracle','leg_seconds','write_phase')} if isinstance(v,dict) else v)
PY
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
timeout 400 rocprofv3 --kernel-trace --stats -d $OUT/k_default -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/bench_default_traced.json 2> $OUT/bench_default_traced.err
$EXP stats $(db $OUT/k_default) $OUT/kernel_stats_bench_default.csv; $EXP trace $(db $OUT/k_default) $OUT/kernel_trace_bench_default.csv
gzip -f $OUT/kernel_trace_bench_default.csv
head -9 $OUT/kernel_stats_bench_default.csv | cut -c1-220
if [ -n "$SKIP_PMC" ]; then find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete; exit 0; fi
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq -o s -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_sq.json 2> $OUT/ring_pmc_sq.err
$EXP counters $(db $OUT/pmc_sq) $OUT/pmc_sq_ring.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_fetch.json 2> $OUT/ring_pmc_fetch.err
$EXP counters $(db $OUT/pmc_fetch) $OUT/pmc_fetch_size_ring.csv
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_write.json 2> $OUT/ring_pmc_write.err
$EXP counters $(db $OUT/pmc_write) $OUT/pmc_write_size_ring.csv
python3 $ROOT/scripts/r6_traffic.py $OUT/pmc_fetch_size_ring.csv $OUT/pmc_write_size_ring.csv $OUT/ring_pmc_fetch.json $OUT/traffic.json
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
grep -i "sha256" $OUT/pmc_sq_ring.csv $OUT/pmc_fetch_size_ring.csv | sed 's/void pbsk:://' | cut -c1-230
cat $OUT/ring_pmc_sq.json | cut -c1-400
# the N-rank code path of bench.py on the one GPU there is: (1) RCCL with ONE rank (process group, barriers, collectives, C-ABI reduce),
# (2) two ranks sharing the GPU over gloo, spawned by bench.py itself (--gpus 2 without a launcher)
cd $ROOT
( PBS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $out/bench_force_dist_nccl_1rank.json 2> $out/bench_force_dist_nccl_1rank.err; echo "force_dist rc=$?" )
( PBS_BENCH_BACKEND=gloo timeout 250 python bench.py --gpus 2 --steps 3 --warmup 1 --arena-gib 96 --ring-sha-cus 64 --no-extras > $out/bench_gloo_2ranks_one_gpu.json 2> $out/bench_gloo_2ranks_one_gpu.err; echo "gloo2 rc=$?" )
python3 - <<PY
import json
for n in ('bench_force_dist_nccl_1rank', 'bench_gloo_2ranks_one_gpu'):
    try:
        for l in open('$out/%s.json' % n):
            if l.startswith('{'):
                d = json.loads(l); print(n, d['value'], 'n_gpus', d['n_gpus'], 'scaling', d.get('scaling'), (d.get('results') or {}).get('c_abi_digest_reduce'), (d.get('cpu_baseline') or {}).get('all_ranks'))
    except Exception as e:
        print(n, 'no line', e)
PY
tail -2 $out/bench_force_dist_nccl_1rank.err $out/bench_gloo_2ranks_one_gpu.err | cut -c1-300
