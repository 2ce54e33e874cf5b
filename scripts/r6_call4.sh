#!/bin/bash
# round 6, call 4: page geometry A/B on configs[2..4] through the ring (ring_manyfiles: half the bytes in 16 MiB chunks — the
# workload whose pages were held for a full chain), same box, 61 tiles vs 8 tiles
out=gpurun_out/r6c4; mkdir -p $out
export PYTHONFAULTHANDLER=1
for t in 61 8 61 8; do
for w in ring_manyfiles ring_corpus_dup; do
  timeout 300 python bench.py --gpus 1 --workload $w --steps 6 --warmup 1 --no-cpu-baseline --ring-page-tiles $t > $out/${w}_t$t.json 2> $out/${w}_t$t.err
  python3 - $out/${w}_t$t.json "$w tiles=$t" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d.get('roofline',{})
        print(sys.argv[2], d['value'], d['ms_per_step'], {k:v for k,v in (r.get('feed_phase') or {}).items() if k!='note'}, d['config'].get('sha_cus'), d['config'].get('express_cus'))
PY
  tail -2 $out/${w}_t$t.err | cut -c1-200
done
done
