#!/bin/bash
# A/B of the ring's backlog gate (PBSGPU_RING_BACKLOG_MIB; 0 = off)   usage: scripts/r3_backlog_ab.sh OUTDIR [MiB|default ...]
out=$1; shift
mkdir -p "$out"
export PBS_BENCH_RING_TRACE=1
for b in "$@"; do
  if [ "$b" = "default" ]; then unset PBSGPU_RING_BACKLOG_MIB; else export PBSGPU_RING_BACKLOG_MIB=$b; fi
  timeout 300 python bench.py --no-extras --no-cpu-baseline > "$out/backlog_$b.json" 2> "$out/backlog_$b.err"
  python - "$out/backlog_$b.json" "$b" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    f = d["roofline"]["feed_phase"]
    print("backlog_mib", sys.argv[2], "value", d["value"], "feed", f["GiBps"], "feed_s", f["seconds"], "drain_s", f["drain_seconds"],
          "rounds", d["config"]["rounds_in_timed_region"], "single_ms", d["roofline"]["single_file"]["ms"])
except Exception as e:
    print("backlog_mib", sys.argv[2], "FAILED", e)
PY
  grep "ring trace" "$out/backlog_$b.err"
done
