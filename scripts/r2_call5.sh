mkdir -p gpurun_out/r2c5
PBSGPU_TRACE=1 python scripts/r2_probe_stream.py 2>&1 | grep -v "^\[pbsgpu\] stream" | tail -8
for p in 1 2 3 4 8; do timeout 300 python bench.py --workload hostfeed --producers $p --steps 8 --warmup 4 > gpurun_out/r2c5/hostfeed_p$p.json 2> gpurun_out/r2c5/hostfeed_p$p.err; tail -2 gpurun_out/r2c5/hostfeed_p$p.err; python -c "
import json; d=json.load(open('gpurun_out/r2c5/hostfeed_p$p.json')); print('hostfeed p=$p', d['value'], d['roofline']['frac_of_measured_h2d'], d['stream_records_match_oracle'])"; done
timeout 600 python -m pytest tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
