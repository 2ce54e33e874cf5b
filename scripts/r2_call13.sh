for iv in 56 30 0; do echo "== interval $iv"; for P in 1 8; do PBSGPU_HASH_INTERVAL_MS=$iv python scripts/r2_probe_feed.py $P 24 2>&1 | tail -1; done; done
for args in "--producers 8 --steps 24" "--producers 8 --steps 24 --tee" "--producers 1 --steps 24"; do
timeout 300 python bench.py --workload hostfeed $args 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); print('hostfeed $args', d['value'], d['roofline']['frac_of_measured_h2d'], d['config']['xxh3_tee_files'], d['stream_records_match_oracle'])"; done
