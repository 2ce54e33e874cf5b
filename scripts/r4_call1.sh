#!/bin/bash
# round 4, GPU call 1: GPU_MAX_HW_QUEUES=20 as the default — full GPU suite + the driver's command
out=gpurun_out/r4c1; mkdir -p $out
( time timeout 600 python -m pytest tests -m gpu -x -q ) > $out/pytest.log 2>&1; tail -3 $out/pytest.log
timeout 400 python bench.py --gpus 1 --steps 20 --warmup 5 > $out/bench.json 2> $out/bench.err; tail -c 3000 $out/bench.json
