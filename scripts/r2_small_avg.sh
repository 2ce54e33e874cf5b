timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu -k "parallel_resolve" 2>&1 | tail -12 | cut -c1-600
for mode in pair lane; do for avg in 65536 262144; do
PBSGPU_SHA_MODE=$mode timeout 300 python bench.py --avg $avg --steps 12 --warmup 2 --cpu-sample-gib 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; print('$mode avg=$avg', d['value'], 'GiB/s', d['ms_per_step'], 'frac_valu', r['frac'], 'sha_ms', k['k_sha256_pair<RecordSource>']['kernel_ms'], 'scan_ms', k['k_scan3<34,4>']['kernel_ms'], 'res', k['resolve_chain'], 'serial', d['serial_step_ms'], d['config']['chunks_per_batch'], d['cpu_baseline']['records_match_gpu'])"
done; done
