mkdir -p gpurun_out/r2c4
PBSGPU_TRACE=1 python scripts/r2_probe_stream.py 2>&1 | grep -v "^\[pbsgpu\] stream" | tail -12
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/r2c4/gpu_tests.log 2>&1; echo "pytest rc=$?"; tail -5 gpurun_out/r2c4/gpu_tests.log
for p in 1 2 4 8; do timeout 300 python bench.py --workload hostfeed --producers $p --steps 4 --warmup 4 > gpurun_out/r2c4/hostfeed_p$p.json 2> gpurun_out/r2c4/hostfeed_p$p.err; python -c "
import json; d=json.load(open('gpurun_out/r2c4/hostfeed_p$p.json')); print('hostfeed p=$p', d['value'], d['roofline']['frac_of_measured_h2d'], d['stream_records_match_oracle'])"; done
for w in stream64g manyfiles; do timeout 300 python bench.py --workload $w --no-cpu-baseline > gpurun_out/r2c4/$w.json 2> gpurun_out/r2c4/$w.err; python -c "
import json; d=json.load(open('gpurun_out/r2c4/$w.json')); print('$w', d['value'], d['ms_per_step'], d['roofline']['latency_bound']['frac_of_bound'], d['serial_step_ms'])"; done
timeout 300 python bench.py --collect fifo --no-cpu-baseline > gpurun_out/r2c4/stream_fifo.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2c4/stream_fifo.json')); print('fifo', d['value'])"
timeout 300 python bench.py --reread 16 --steps 32 --no-cpu-baseline > gpurun_out/r2c4/reread16.json 2>/dev/null; python -c "
import json; d=json.load(open('gpurun_out/r2c4/reread16.json')); print('reread16', d['value'])"
