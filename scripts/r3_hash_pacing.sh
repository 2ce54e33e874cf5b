#!/bin/bash
# Launch pacing of the streams' shared hash jobs: lanes x interval (0 = greedy)   usage: r3_hash_pacing.sh OUTDIR "lanes:interval_ms ..."
out=$1; mkdir -p "$out"
for li in $2; do
  lanes=${li%%:*}; iv=${li##*:}
  for P in 1 8; do
    steps=96; [ $P = 8 ] && steps=32
    PBSGPU_HASH_LANES=$lanes PBSGPU_HASH_INTERVAL_MS=$iv timeout 200 python bench.py --workload hostfeed --producers $P --steps $steps --warmup 4 > "$out/hf_p${P}_l${lanes}_iv$iv.json" 2> "$out/hf_p${P}_l${lanes}_iv$iv.err"
    python - "$out/hf_p${P}_l${lanes}_iv$iv.json" $P $lanes $iv <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    w = d["write_phase"]
    print("writers", sys.argv[2], "lanes", sys.argv[3], "interval_ms", sys.argv[4], "value", d["value"], "write_phase", w["GiBps"], "drain_s", w["drain_seconds"], "match", d["stream_records_match_oracle"])
except Exception as e:
    print("writers", sys.argv[2], "lanes", sys.argv[3], "interval_ms", sys.argv[4], "FAILED", e)
PY
  done
done
