# dense-regime A/B: four producer/consumer pairs per CU vs the single-wave kernel at 2 / 4 waves per SIMD
timeout 1500 python -m pytest tests -x -q -m gpu 2>&1 | tail -3 | cut -c1-600
run() { env "$@" timeout 300 python bench.py --avg $AVG --steps 10 --warmup 2 --cpu-sample-gib 1 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('$* avg=$AVG', d['value'], 'GiB/s', d['ms_per_step'], 'serial', d['serial_step_ms'], d['cpu_baseline']['records_match_gpu'])"; }
for AVG in 65536 262144; do
run PBSGPU_SHA_DENSE_FORM=pairs
run PBSGPU_SHA_DENSE_FORM=lanes PBSGPU_SHA_LANE_WAVES=2
run PBSGPU_SHA_DENSE_FORM=lanes PBSGPU_SHA_LANE_WAVES=4
done
timeout 300 python bench.py 2>/dev/null | tee gpurun_out/bench_hostdense_default.json | python -c "
import json,sys; d=json.load(sys.stdin); print('default', d['value'], d['ms_per_step'], d['serial_step_ms'], d['roofline']['latency_bound']['frac_of_bound'], d['cpu_baseline']['records_match_gpu'])"
