timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu > gpurun_out/gpu_tests_c22.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Thread\|^$" gpurun_out/gpu_tests_c22.log | head -40 | cut -c1-300; tail -3 gpurun_out/gpu_tests_c22.log
for rep in a b; do for v in par serial; do
if [ $v = serial ]; then export PBSGPU_RESOLVE_SERIAL=1; else unset PBSGPU_RESOLVE_SERIAL; fi
timeout 300 python bench.py --steps 24 --cpu-sample-gib 4 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; print('$v', d['value'], d['ms_per_step'], 'res', k['resolve_chain'], 'serial', d['serial_step_ms'], d['cpu_baseline']['records_match_gpu'], d['cpu_baseline']['records_checked'])"
done; done
