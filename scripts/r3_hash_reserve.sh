#!/bin/bash
# Reserve hash lanes (taken only by launches somebody blocks on)   usage: r3_hash_reserve.sh OUTDIR "reserve ..." 
out=$1; mkdir -p "$out"; n=0
for rs in $2; do
  n=$((n+1))
  for P in 1 8; do
    steps=96; [ $P = 8 ] && steps=32
    PBSGPU_HASH_RESERVE_LANES=$rs timeout 200 python bench.py --workload hostfeed --producers $P --steps $steps --warmup 4 > "$out/hf_p${P}_r${rs}_$n.json" 2> "$out/hf_p${P}_r${rs}_$n.err"
    python - "$out/hf_p${P}_r${rs}_$n.json" $P $rs <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    w = d["write_phase"]
    print("writers", sys.argv[2], "reserve_lanes", sys.argv[3], "value", d["value"], "write_phase", w["GiBps"], "drain_s", w["drain_seconds"], "match", d["stream_records_match_oracle"])
except Exception as e:
    print("writers", sys.argv[2], "reserve_lanes", sys.argv[3], "FAILED", e)
PY
  done
done
