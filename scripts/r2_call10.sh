timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_c10.log 2>&1; echo "pytest rc=$?"; head -30 gpurun_out/gpu_tests_c10.log | cut -c1-300; tail -4 gpurun_out/gpu_tests_c10.log
bash scripts/r2_verify64.sh
timeout 600 python bench.py --workload verify --steps 5 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['results']; print({k:r[k] for k in r if k not in ('note',)})"
