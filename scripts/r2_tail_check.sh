# parity first (everything that hashes), then the small-average regime in both SHA modes, then the default line
timeout 1200 python -m pytest tests -x -q -m gpu 2>&1 | tail -4 | cut -c1-400
for mode in pair lane; do for avg in 65536 262144; do
PBSGPU_SHA_MODE=$mode timeout 300 python bench.py --avg $avg --steps 12 --warmup 2 --cpu-sample-gib 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; sk=[x for x in k if x.startswith('k_sha256')][0]; print('$mode avg=$avg', d['value'], 'GiB/s', d['ms_per_step'], 'frac_valu', r['frac'], 'sha_ms', k[sk]['kernel_ms'], 'res', k['resolve_chain'], 'serial', d['serial_step_ms'], d['config']['chunks_per_batch'], d['cpu_baseline']['records_match_gpu'])"
done; done
timeout 300 python bench.py 2>/dev/null | tee gpurun_out/bench_tail_default.json | cut -c1-300
