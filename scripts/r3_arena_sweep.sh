#!/bin/bash
# Arena-size sweep of the ring workload: does a smaller arena (= less unhashed backlog at the end of the feed phase) shorten
# the drain without lowering the feed rate?   usage: scripts/r3_arena_sweep.sh OUTDIR [arena GiB ...]
out=$1; shift
mkdir -p "$out"
export PBS_BENCH_RING_TRACE=1
for a in "$@"; do
  extra=""
  [ "$a" != "default" ] && extra="--arena-gib $a"
  timeout 300 python bench.py --no-extras --no-cpu-baseline $extra > "$out/arena_$a.json" 2> "$out/arena_$a.err"
  python - "$out/arena_$a.json" "$a" <<'PY'
import json, sys
try:
    d = json.loads([l for l in open(sys.argv[1]) if l.startswith("{")][0])
    f = d["roofline"]["feed_phase"]
    print("arena", sys.argv[2], "value", d["value"], "feed", f["GiBps"], "feed_s", f["seconds"], "drain_s", f["drain_seconds"],
          "pages", d["config"]["arena_pages"], "rounds", d["config"]["rounds_in_timed_region"], "single_ms", d["roofline"]["single_file"]["ms"])
except Exception as e:
    print("arena", sys.argv[2], "FAILED", e)
PY
  grep "ring trace" "$out/arena_$a.err"
done
