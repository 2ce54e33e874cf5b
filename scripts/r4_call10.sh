#!/bin/bash
# round 4, GPU call 10: where does the chain's time go? batch path (one 64 GiB batch: the SHA launch IS the chain of its longest chunk)
# with the pair kernel and with the express kernel; the ring's lone file at 192 / 64 service CUs; SQ counters of both batch launches
out=gpurun_out/r4c10; mkdir -p $out
export PYTHONFAULTHANDLER=1
show() { python3 - "$1" "$2" <<'PY'
import json,sys
d=json.loads([l for l in open(sys.argv[1]) if l.startswith('{')][0]); r=d['roofline']
keys=[k for k in r if k in ('kernels','uncontended','single_file','serial_step_ms','latency_bound','feed_phase')]
print(sys.argv[2], d['value'], {k:r[k] for k in keys}, d.get('serial_step_ms'))
PY
}
timeout 200 python bench.py --workload stream64g --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/batch_pair.json 2>$out/batch_pair.err; show $out/batch_pair.json batch_pair
PBSGPU_SHA_MODE=xpair timeout 200 python bench.py --workload stream64g --steps 2 --warmup 1 --no-extras --no-cpu-baseline > $out/batch_xpair.json 2>$out/batch_xpair.err; show $out/batch_xpair.json batch_xpair
timeout 200 python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $out/ring_lone_192.json 2>/dev/null; show $out/ring_lone_192.json ring_lone_192
PBSGPU_RING_SHA_CUS=64 timeout 200 python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $out/ring_lone_64.json 2>/dev/null; show $out/ring_lone_64.json ring_lone_64
PBSGPU_RING_XP_CUS=32 PBSGPU_RING_LONG_BYTES=1048576 timeout 200 python bench.py --gpus 1 --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $out/ring_lone_xp_all.json 2>/dev/null; show $out/ring_lone_xp_all.json ring_lone_xp32_everything_express
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
for mode in pair xpair; do
  PBSGPU_SHA_MODE=$mode timeout 300 rocprofv3 --pmc GRBM_GUI_ACTIVE SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_INSTS_SALU SQ_INSTS_LDS -d $OUT/pmc_$mode -o s -- python3 $ROOT/bench.py --workload stream64g --steps 1 --warmup 0 --no-extras --no-cpu-baseline > $OUT/pmc_$mode.json 2> $OUT/pmc_$mode.err
  python3 $ROOT/scripts/rocpd_export.py counters $(find $OUT/pmc_$mode -name "*_results.db" | head -1) $OUT/pmc_sq_batch_$mode.csv
  grep "sha256" $OUT/pmc_sq_batch_$mode.csv | sed 's/"void pbsk::k_sha256_[a-z]*<[^"]*"/SHA/' | cut -c1-150
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
