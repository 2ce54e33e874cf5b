#!/bin/bash
# round 6, call 10: the multi-threaded stream test that hung in the full suite: alone; and with a huge graveyard cap
out=gpurun_out/r6c10; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 200 python -m pytest tests/test_gpu_round2.py -m gpu -q --timeout 90 -x -k "many_streams_and_batches" ) > $out/alone.log 2>&1; grep -a "passed\|failed\|Timeout" $out/alone.log | tail -3
( time PBSGPU_GRAVEYARD_MIB=100000000 timeout 200 python -m pytest tests/test_gpu_round2.py -m gpu -q --timeout 90 -x -k "many_streams_and_batches" ) > $out/bigcap.log 2>&1; grep -a "passed\|failed\|Timeout" $out/bigcap.log | tail -3
( time timeout 400 python -m pytest tests/test_gpu_round2.py -m gpu -q --timeout 120 -x ) > $out/round2.log 2>&1; grep -a "passed\|failed\|Timeout" $out/round2.log | tail -3
