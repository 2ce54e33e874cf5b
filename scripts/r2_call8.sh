mkdir -p gpurun_out/r2c8
for P in 1 2 4 8; do python scripts/r2_probe_feed.py $P 16 2>&1 | tail -1; done
for P in 4 8; do python scripts/r2_probe_feed.py $P 48 2>&1 | tail -1; done
timeout 1500 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3
for p in 1 2 4 8; do timeout 300 python bench.py --workload hostfeed --producers $p > gpurun_out/r2c8/hostfeed_p$p.json 2> gpurun_out/r2c8/hostfeed_p$p.err; tail -2 gpurun_out/r2c8/hostfeed_p$p.err; python -c "
import json; d=json.load(open('gpurun_out/r2c8/hostfeed_p$p.json')); print('hostfeed p=$p', d['value'], d['roofline']['frac_of_measured_h2d'], d['stream_records_match_oracle'])"; done
