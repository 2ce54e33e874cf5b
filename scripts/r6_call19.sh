#!/bin/bash
# round 6, call 19: the two-ring graveyard test, alone, three times; then the rest of the suite behind it
out=gpurun_out/r6c19; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in 1 2 3; do
  timeout 600 python -m pytest tests/test_gpu_round5.py -m gpu -q -k two_busy_rings > $out/two_rings_$i.log 2>&1; tail -1 $out/two_rings_$i.log
done
timeout 1500 python -m pytest tests -m gpu -q > $out/pytest_gpu.log 2>&1; tail -3 $out/pytest_gpu.log
