for rep in 1 2; do for args in "--producers 8" "--producers 8 --steps 48"; do
timeout 300 python bench.py --workload hostfeed $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed $args', d['value'], d['roofline']['frac_of_measured_h2d'], d['stream_records_match_oracle'])"; done; done
for args in "--producers 8 --steps 48 --tee" "--producers 4 --steps 48" "--producers 1 --steps 24"; do
timeout 300 python bench.py --workload hostfeed $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed $args', d['value'], d['roofline']['frac_of_measured_h2d'], d['config']['xxh3_tee_files'], d['stream_records_match_oracle'])"; done
