#!/bin/bash
# round 6, call 5: what clock64 counts; the driver's line with the regime probe (twice); ring tests with the probe in the kernels
out=gpurun_out/r6c5; mkdir -p $out
export PYTHONFAULTHANDLER=1
./scripts/ubench/clk > $out/ubench_clk.log 2>&1; cat $out/ubench_clk.log
rocm-smi --showclocks 2>/dev/null | grep -i "sclk\|mclk\|fclk" | head -6
for i in 1 2; do
timeout 500 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_default_$i.json 2> $out/bench_default_$i.err
python3 - $out/bench_default_$i.json <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']
        print('default', d['value'], d['ms_per_step'], {k:v for k,v in r.get('feed_phase',{}).items() if k!='note'}, {k:v for k,v in r['single_file'].items() if k!='note'})
        print('   regime', json.dumps({k:v for k,v in r['regime'].items() if k!='note'}))
PY
tail -2 $out/bench_default_$i.err | cut -c1-300
done
( time timeout 600 python -m pytest tests/test_gpu_ring.py tests/test_gpu_xpair.py -m gpu -q --timeout 300 -x ) > $out/pytest.log 2>&1; tail -4 $out/pytest.log | cut -c1-400
