set -x
mkdir -p gpurun_out/r2c1
run() { tag=$1; shift; env "$@" > gpurun_out/r2c1/$tag.json 2> gpurun_out/r2c1/$tag.err; tail -c 400 gpurun_out/r2c1/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2c1/$tag.json")); print("$tag", d["value"], d["ms_per_step"], d["roofline"]["kernel_ms"], d["roofline"]["scan_kernel"]["kernel_ms"], d["serial_value"])
except Exception as e: print("$tag FAILED", e)
PY
}
run pair_if4  PBSGPU_SHA_MODE=pair timeout 300 python bench.py --no-cpu-baseline --inflight 4 --steps 16
run lane_if4  PBSGPU_SHA_MODE=lane timeout 300 python bench.py --no-cpu-baseline --inflight 4 --steps 16
run lane_if8  PBSGPU_SHA_MODE=lane timeout 300 python bench.py --no-cpu-baseline --inflight 8 --steps 24
run lane_if16 PBSGPU_SHA_MODE=lane timeout 300 python bench.py --no-cpu-baseline --inflight 16 --steps 32
run lane_if16_any PBSGPU_SHA_MODE=lane timeout 300 python bench.py --no-cpu-baseline --inflight 16 --steps 32 --collect any
run pair_if16_any PBSGPU_SHA_MODE=pair timeout 300 python bench.py --no-cpu-baseline --inflight 16 --steps 32 --collect any
run lane_if16_pad18 PBSGPU_SHA_MODE=lane PBSGPU_SHA_LDS_PAD=18000 timeout 300 python bench.py --no-cpu-baseline --inflight 16 --steps 32
