#!/bin/bash
# round 5, call 10: kernel traces with the staged round tables — the default split, and 184 + 16 CUs (every round full)
out=gpurun_out/r5c10; mkdir -p $out
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
for cfg in default pair184_xp16; do
  e="A=1"; [ $cfg = pair184_xp16 ] && e="PBSGPU_RING_SHA_CUS=184 PBSGPU_RING_XP_CUS=16"
  env $e timeout 300 rocprofv3 --kernel-trace --stats -d $OUT/k_$cfg -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $OUT/bench_$cfg.json 2> $OUT/bench_$cfg.err
  $EXP stats $(db $OUT/k_$cfg) $OUT/kernel_stats_$cfg.csv; $EXP trace $(db $OUT/k_$cfg) $OUT/kernel_trace_$cfg.csv
  gzip -f $OUT/kernel_trace_$cfg.csv
  head -9 $OUT/kernel_stats_$cfg.csv | cut -c1-200
  grep -o '"value": [0-9.]*' $OUT/bench_$cfg.json | head -1
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
