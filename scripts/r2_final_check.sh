timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3
python3 bench.py --gpus 1 --steps 20 --warmup 5 > gpurun_out/final_bench.json 2> gpurun_out/final_bench.err; python -c "
import json; d=json.load(open('gpurun_out/final_bench.json')); print('default', d['value'], d['ms_per_step'], d['roofline']['frac'], d['roofline']['latency_bound']['frac_of_bound'], d['serial_value'], d['cpu_baseline']['records_match_gpu'])"
