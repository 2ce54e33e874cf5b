#!/bin/bash
# round 4, GPU call 2: the ring as the one engine — ring tests, the stream surface on the ring, smoke, the default line (fused control kernel)
out=gpurun_out/r4c2; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_ring.py -x -q --timeout 240 ) > $out/pytest_ring.log 2>&1; tail -25 $out/pytest_ring.log
( time timeout 900 python -m pytest tests/test_gpu_parity.py tests/test_gpu_round2.py tests/test_gpu_round3.py -m gpu -x -q --timeout 300 \
   -k "stream or payload or tee or archive or entry or suggest or writer or engine_may or many_streams or helper or random_programs or cpp_mirror" ) > $out/pytest_streams.log 2>&1; tail -25 $out/pytest_streams.log
( timeout 200 python __graft_entry__.py smoke ) > $out/smoke.log 2>&1; tail -3 $out/smoke.log
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $out/bench_fused.json 2> $out/bench_fused.err; tail -c 1500 $out/bench_fused.json; tail -5 $out/bench_fused.err
PBSGPU_RING_FUSED=0 timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_unfused.json 2> $out/bench_unfused.err; tail -c 600 $out/bench_unfused.json
timeout 200 python bench.py --workload hostfeed --producers 1 --steps 96 --warmup 8 > $out/hf1.json 2> $out/hf1.err; tail -c 1200 $out/hf1.json; tail -3 $out/hf1.err
