#!/bin/bash
# round 5, call 13: the four GPU tests added after the final evidence run (round tables staged / from host memory; light-load express rule on / off)
out=gpurun_out/r5c13; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_round5.py -q --timeout 300 -k "staged_or_read or longer_chunks_express" -v ) > $out/pytest.log 2>&1; grep -a "PASSED\|FAILED\|passed\|failed\|Error\|assert" $out/pytest.log | tail -12 | cut -c1-300
