python scripts/r2_probe_smallwrites.py 2>&1 | tail -3
timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3
timeout 300 python bench.py --workload hostfeed --producers 8 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed p8', d['value'], d['roofline']['frac_of_measured_h2d'], d['stream_records_match_oracle'], d['cpu_baseline']['value'])"
