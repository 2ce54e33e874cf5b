mkdir -p gpurun_out/r2c18
run() { tag=$1; shift; env "$@" > gpurun_out/r2c18/$tag.json 2> gpurun_out/r2c18/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2c18/$tag.json")); r=d["roofline"]; k=r.get("kernels",{})
    print("$tag", d["value"], d["ms_per_step"], "sha", k.get("k_sha256_pair<RecordSource>",{}).get("kernel_ms"), "scan", k.get("k_scan3<34,4>",{}).get("kernel_ms"), "res", k.get("resolve_chain",{}).get("kernel_ms"), "serial", d.get("serial_step_ms"))
except Exception as e: print("$tag FAILED", e)
PY
}
for rep in a b; do
run tpw4_$rep PBSGPU_SCAN_TILES_PER_WAVE=4 timeout 300 python bench.py --no-cpu-baseline
run tpw0_$rep PBSGPU_SCAN_TILES_PER_WAVE=0 timeout 300 python bench.py --no-cpu-baseline
run tpw1_$rep PBSGPU_SCAN_TILES_PER_WAVE=1 timeout 300 python bench.py --no-cpu-baseline
run tpw16_$rep PBSGPU_SCAN_TILES_PER_WAVE=16 timeout 300 python bench.py --no-cpu-baseline
done
run many_tpw4 PBSGPU_SCAN_TILES_PER_WAVE=4 timeout 300 python bench.py --workload manyfiles --no-cpu-baseline
timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu 2>&1 | tail -3
