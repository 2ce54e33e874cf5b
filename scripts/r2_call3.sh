set -x
mkdir -p gpurun_out/r2c3
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/r2c3/gpu_tests.log 2>&1; echo "pytest rc=$?"
head -40 gpurun_out/r2c3/gpu_tests.log; tail -30 gpurun_out/r2c3/gpu_tests.log
for p in 1 2 4 8; do timeout 300 python -X faulthandler bench.py --workload hostfeed --producers $p --steps 4 --warmup 4 > gpurun_out/r2c3/hostfeed_p$p.json 2> gpurun_out/r2c3/hostfeed_p$p.err; head -30 gpurun_out/r2c3/hostfeed_p$p.err; cat gpurun_out/r2c3/hostfeed_p$p.json; done
