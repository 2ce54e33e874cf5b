#!/bin/bash
# round 6, call 11: the whole -m gpu suite again (grace period once per park request)
out=gpurun_out/r6c11; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 1000 python -m pytest tests -m gpu -q --timeout 200 -x --durations=12 ) > $out/pytest.log 2>&1; grep -a "passed\|failed\|FAILED\|Error\|Timeout" $out/pytest.log | tail -8 | cut -c1-300; grep -a -A14 "slowest" $out/pytest.log | cut -c1-160
