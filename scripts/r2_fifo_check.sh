timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | cut -c1-300
PBSGPU_SHA_FIFO=4 timeout 240 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -2 | cut -c1-300
for fifo in 2 4; do for pct in 150 0; do for avg in 65536 262144; do
PBSGPU_SHA_FIFO=$fifo PBSGPU_SHA_DENSE_PCT=$pct timeout 300 python bench.py --avg $avg --steps 10 --warmup 2 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; sk=[x for x in k if x.startswith('k_sha256')][0]; print('fifo=$fifo dense_pct=$pct avg=$avg', d['value'], 'GiB/s', d['ms_per_step'], 'sha_ms', k[sk]['kernel_ms'], 'serial', d['serial_step_ms'])"
done; done; done
for fifo in 2 4; do
PBSGPU_SHA_FIFO=$fifo timeout 300 python bench.py --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('default fifo=$fifo', d['value'], d['ms_per_step'], d['serial_step_ms'])"
done
