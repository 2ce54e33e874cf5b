timeout 1800 python -X faulthandler -m pytest tests/test_gpu_round2.py -x -q -m gpu > gpurun_out/gpu_tests_c11.log 2>&1; echo "pytest rc=$?"; grep -v "^  File\|^Thread\|^$" gpurun_out/gpu_tests_c11.log | head -60 | cut -c1-400; tail -4 gpurun_out/gpu_tests_c11.log
for P in 8; do timeout 300 python bench.py --workload hostfeed --producers $P --tee --steps 24 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed tee', d['value'], d['roofline']['frac_of_measured_h2d'], d['config']['xxh3_tee_files'], d['stream_records_match_oracle'])"; done
timeout 300 python bench.py --workload hostfeed --producers 1 --tee --steps 24 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed tee p1', d['value'], d['roofline']['frac_of_measured_h2d'], d['config']['xxh3_tee_files'])"
timeout 300 python bench.py --workload hostfeed --producers 1 --steps 24 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed p1', d['value'], d['roofline']['frac_of_measured_h2d'])"
