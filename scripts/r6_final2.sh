#!/bin/bash
# round 6, second session: soak of the ring tests on the final code (queue positions by ticket, probe counters through the status
# block), then scripts/r6_final.sh (the whole suite, the driver's line with every leg, rocprofv3 kernel stats + PMC passes, N-rank legs)
out=gpurun_out/r6soak; mkdir -p $out
export PYTHONFAULTHANDLER=1
for i in 1 2 3 4; do
  ( timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_dense.py tests/test_gpu_round4.py tests/test_gpu_round5.py -m gpu -q -x ) > $out/soak_$i.log 2>&1; echo "soak $i: $(tail -1 $out/soak_$i.log | cut -c1-120)"
done
( PBSGPU_RING_LANES_CUS=2 timeout 900 python -m pytest tests/test_gpu_ring.py tests/test_gpu_dense.py -m gpu -q -x ) > $out/soak_lanes.log 2>&1; echo "soak lanes: $(tail -1 $out/soak_lanes.log | cut -c1-120)"
bash scripts/r6_final.sh
