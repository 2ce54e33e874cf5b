#!/bin/bash
# round 4, GPU call 6: why ncclCommInitRank fails behind the C ABI (one rank), rest of the round-4 tests, single file with a whole round per turn
out=gpurun_out/r4c6; mkdir -p $out
export PYTHONFAULTHANDLER=1
env | grep -i "nccl\|rccl\|HSA_\|LD_LIBRARY" > $out/env.txt
ip -o addr 2>/dev/null | cut -c1-120 > $out/ifaces.txt; cat /proc/net/dev | cut -c1-60 >> $out/ifaces.txt
NCCL_DEBUG=INFO NCCL_DEBUG_SUBSYS=INIT,BOOTSTRAP,NET,ENV PBSGPU_TRACE=1 timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --timeout 200 -k "comm" > $out/comm_default.log 2>&1; tail -5 $out/comm_default.log | cut -c1-300
grep -i "warn\|error\|fail" $out/comm_default.log | head -20 | cut -c1-300
NCCL_SOCKET_IFNAME=lo NCCL_DEBUG=WARN timeout 300 python -m pytest tests/test_gpu_round4.py -m gpu -q --timeout 200 -k "comm" > $out/comm_lo.log 2>&1; tail -3 $out/comm_lo.log | cut -c1-300
timeout 120 python - > $out/comm_torch_first.log 2>&1 <<'PY'
import torch, numpy as np, os
import pbs_plus_amd
from pbs_plus_amd import Comm, buzhash
eng = pbs_plus_amd.Engine(buzhash.NewConfig(4096), device=0, inflight=1)
try:
    c = Comm(eng, Comm.unique_id(), 0, 1); print("torch imported first: comm ok")
except Exception as e:
    print("torch imported first:", e)
print([l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip64' in l][:0] or sorted({l.split()[-1] for l in open('/proc/self/maps') if 'rccl' in l or 'amdhip64' in l}))
PY
cat $out/comm_torch_first.log | tail -4 | cut -c1-400
( time timeout 600 python -m pytest tests/test_gpu_round4.py -m gpu -q --timeout 300 ) > $out/pytest_round4.log 2>&1; tail -6 $out/pytest_round4.log | cut -c1-250
timeout 200 python bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $out/bench_single.json 2>/dev/null
python3 -c "
import json
d=json.loads([l for l in open('$out/bench_single.json') if l.startswith('{')][0]); r=d['roofline']; print('4 steps:', d['value'], 'single file', r['single_file'])"
PBSGPU_RING_LONE_DEFER_MS=0 timeout 200 python bench.py --gpus 1 --steps 4 --warmup 1 --no-extras --no-cpu-baseline > $out/bench_single_nodefer.json 2>/dev/null
python3 -c "
import json
d=json.loads([l for l in open('$out/bench_single_nodefer.json') if l.startswith('{')][0]); r=d['roofline']; print('4 steps, no lone deferral:', d['value'], 'single file', r['single_file'])"
