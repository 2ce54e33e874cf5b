#!/bin/bash
# Is the slowdown behind a host-fed leg a property of the PROCESS or of the DEVICE? Process A: eight writers x 32 GiB, exits.
# Process B (started at once): the batch workload. rocm-smi clocks / power / temperature in between.
out=${1:-gpurun_out/r3ag}; mkdir -p $out
smi() { rocm-smi --showclocks --showpower --showtemp 2>/dev/null | grep -E "sclk|mclk|fclk|Power|Temperature \(Sensor (edge|junction|memory)" | tr -s ' ' | head -8 | sed "s/^/[$1] /"; }
val() { python3 -c "
import json,sys
for l in open('$1'):
    if l.startswith('{'):
        d=json.loads(l); k=d['roofline'].get('kernels') or {}
        print('$2', d['value'], 'GiB/s', {n: v.get('kernel_ms') for n, v in k.items() if isinstance(v, dict) and 'sha256' in n})"; }
smi idle
timeout 150 python bench.py --workload stream64g --steps 8 --warmup 4 --no-extras --no-cpu-baseline > $out/batch_before.json 2>/dev/null; val $out/batch_before.json "batch path, fresh process, before:"
timeout 200 python bench.py --workload hostfeed --producers 8 --steps 32 --warmup 4 > $out/hf8.json 2>/dev/null; python3 -c "
import json
for l in open('$out/hf8.json'):
    if l.startswith('{'): print('host-fed 8 writers:', json.loads(l)['value'], 'GiB/s')"
smi after_hostfeed
timeout 150 python bench.py --workload stream64g --steps 8 --warmup 4 --no-extras --no-cpu-baseline > $out/batch_after.json 2>/dev/null; val $out/batch_after.json "batch path, NEW process right after the host-fed process:"
smi after_batch
