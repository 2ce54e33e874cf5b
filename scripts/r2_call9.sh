mkdir -p gpurun_out/r2c9
run() { tag=$1; shift; env "$@" > gpurun_out/r2c9/$tag.json 2> gpurun_out/r2c9/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2c9/$tag.json")); r=d["roofline"]; k=r.get("kernels",{})
    print("$tag", d["value"], d["ms_per_step"], "sha", k.get("k_sha256_pair<RecordSource>",{}).get("kernel_ms"), "scan", k.get("k_scan3<34,4>",{}).get("kernel_ms"), "res", k.get("resolve_chain",{}).get("kernel_ms"), "serial", d.get("serial_step_ms"))
except Exception as e: print("$tag FAILED", e)
PY
}
run res16_a PBSGPU_SCAN_CU_RESERVE=16 timeout 300 python bench.py --no-cpu-baseline
run res0_a PBSGPU_SCAN_CU_RESERVE=0 timeout 300 python bench.py --no-cpu-baseline
run res32_a PBSGPU_SCAN_CU_RESERVE=32 timeout 300 python bench.py --no-cpu-baseline
run res16_b PBSGPU_SCAN_CU_RESERVE=16 timeout 300 python bench.py --no-cpu-baseline
run res0_b PBSGPU_SCAN_CU_RESERVE=0 timeout 300 python bench.py --no-cpu-baseline
run res8_a PBSGPU_SCAN_CU_RESERVE=8 timeout 300 python bench.py --no-cpu-baseline
run many16 PBSGPU_SCAN_CU_RESERVE=16 timeout 300 python bench.py --workload manyfiles --no-cpu-baseline
timeout 600 python bench.py --workload verify --steps 5 > gpurun_out/r2c9/verify.json 2> gpurun_out/r2c9/verify.err; tail -2 gpurun_out/r2c9/verify.err; cat gpurun_out/r2c9/verify.json | cut -c1-3000
timeout 900 python -m pytest tests/test_gpu_round2.py -x -q -m gpu 2>&1 | tail -3
