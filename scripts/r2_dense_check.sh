# parity (everything hashes), then sparse/dense A/B at small averages, then the default line twice
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-400
for pct in 150 0; do for avg in 65536 262144; do
PBSGPU_SHA_DENSE_PCT=$pct timeout 300 python bench.py --avg $avg --steps 12 --warmup 2 --cpu-sample-gib 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; sk=[x for x in k if x.startswith('k_sha256')][0]; print('dense_pct=$pct avg=$avg', d['value'], 'GiB/s', d['ms_per_step'], 'frac_valu', r['frac'], 'sha_ms', k[sk]['kernel_ms'], 'serial', d['serial_step_ms'], d['config']['chunks_per_batch'], d['cpu_baseline']['records_match_gpu'])"
done; done
for i in 1 2; do timeout 300 python bench.py 2>/dev/null | tee gpurun_out/bench_dense_default_$i.json | python -c "
import json,sys; d=json.load(sys.stdin); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['serial_step_ms'], d['roofline']['latency_bound']['frac_of_bound'])"; done
