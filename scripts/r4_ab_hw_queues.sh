#!/bin/bash
# First GPU call of the next round: validate GPU_MAX_HW_QUEUES=20 (and the opt-in shared cut streams) before they become defaults.
#   usage (through gpurun): scripts/r4_ab_hw_queues.sh OUTDIR
# Background: profiles/r03_extras_leg_order.log - a process that has used more than ~20 hardware queues runs every later kernel 19 % slower.
out=${1:-gpurun_out/r4hwq}; mkdir -p $out
line() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); w=d.get('workloads') or {}
        print('$2', d['value'], {k: v.get('value') for k, v in w.items() if isinstance(v, dict)})"; }
for cfg in "24 0" "20 0" "20 4"; do
  set -- $cfg
  export GPU_MAX_HW_QUEUES=$1 PBSGPU_SHARED_CUT_STREAMS=$2
  # the stream writer's parity tests with this configuration
  timeout 400 python -m pytest tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q -k "stream or archive or tee or entry or writer" 2>&1 | tail -1 | sed "s/^/[queues $1 shared $2] /"
  # the driver's command: does the eight-writer leg (last) still lose against a process of its own?
  timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/default_q$1_s$2.json 2> $out/default_q$1_s$2.err; line $out/default_q$1_s$2.json "[queues $1 shared $2] default line:"
  timeout 200 python bench.py --workload hostfeed --producers 8 --steps 32 --warmup 4 > $out/hf8_q$1_s$2.json 2>/dev/null; line $out/hf8_q$1_s$2.json "[queues $1 shared $2] 8 writers, own process:"
  PROBE_MIB=4096 timeout 120 python scripts/r3_probe_residue_chain.py 2>&1 | tail -3 | sed "s/^/[queues $1 shared $2] /"
done
