#!/bin/bash
# round 5, call 2: the adaptive service split (test + configs[2] through the ring with the split chosen by the ring itself, and
# fixed larger express shares beside it), and the chip's clock / power while the driver's line runs
out=gpurun_out/r5c2; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_round5.py -q -x --timeout 400 -k "split or quiesce" -s ) > $out/pytest.log 2>&1
grep -a "passed\|failed\|FAILED\|Error" $out/pytest.log | tail -8 | cut -c1-300
show() { python3 - <<PY
import json
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); print('$2', d['value'], d['roofline'].get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
PY
}
timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_auto.json 2> $out/rmf_auto.err; show $out/rmf_auto.json "ring_manyfiles auto"
timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 2 --no-cpu-baseline > $out/rmf_auto_w2.json 2> $out/rmf_auto_w2.err; show $out/rmf_auto_w2.json "ring_manyfiles auto warmup2"
for xp in 128 144 160; do
  PBSGPU_RING_XP_CUS=$xp timeout 200 python bench.py --workload ring_manyfiles --steps 6 --warmup 1 --no-cpu-baseline > $out/rmf_xp$xp.json 2> $out/rmf_xp$xp.err; show $out/rmf_xp$xp.json "ring_manyfiles xp=$xp"
done
timeout 200 python bench.py --workload ring_corpus_dup --steps 6 --warmup 1 --no-cpu-baseline > $out/rcd_auto.json 2> $out/rcd_auto.err; show $out/rcd_auto.json "ring_corpus_dup auto"
# clock / power while the default line runs (no extras)
( while true; do rocm-smi --showclocks --showpower --showtemp --json 2>/dev/null | tr -d '\n'; echo; sleep 0.25; done ) > $out/smi.log 2>&1 &
SMI=$!
( time timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline ) > $out/bench_noextras.json 2> $out/bench_noextras.err
kill $SMI
python3 - <<PY
import json
for l in open('$out/bench_noextras.json'):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], r.get('feed_phase'), d['config'].get('sha_service_cus'), d['config'].get('express_cus'))
rows=[]
for l in open('$out/smi.log'):
    try: j=json.loads(l)
    except Exception: continue
    c=j.get('card0',{})
    rows.append({k:v for k,v in c.items() if any(t in k.lower() for t in ('sclk','power','mclk','junction','fclk'))})
print(len(rows),'smi samples'); 
for r in rows[::4][:60]: print(r)
PY
