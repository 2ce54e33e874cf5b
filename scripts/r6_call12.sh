#!/bin/bash
# round 6, call 12: the N-rank branch with the C ABI's reduce inside the timed region (one rank over RCCL), the communicator with
# device-resident records
out=gpurun_out/r6c12; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_round2.py -m gpu -q --timeout 300 -x -k "comm or forced_dist or rccl" ) > $out/pytest.log 2>&1; tail -5 $out/pytest.log | cut -c1-300
( PBS_BENCH_FORCE_DIST=1 timeout 300 python bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $out/bench_force_dist_nccl_1rank.json 2> $out/bench_force_dist_nccl_1rank.err; echo "force_dist rc=$?" )
python3 - <<PY
import json
for l in open('$out/bench_force_dist_nccl_1rank.json'):
    if l.startswith('{'):
        d=json.loads(l); print(d['value'], d['n_gpus'], json.dumps((d.get('results') or {}).get('c_abi_digest_reduce'))[:700])
PY
tail -2 $out/bench_force_dist_nccl_1rank.err | cut -c1-300
