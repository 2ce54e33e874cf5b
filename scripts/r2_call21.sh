mkdir -p gpurun_out/r2c21
run() { tag=$1; shift; env "$@" > gpurun_out/r2c21/$tag.json 2> gpurun_out/r2c21/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2c21/$tag.json")); r=d["roofline"]; k=r.get("kernels",{})
    print("$tag", d["value"], d["ms_per_step"], "sha", k.get("k_sha256_pair<RecordSource>",{}).get("kernel_ms"), "scan", k.get("k_scan3<34,4>",{}).get("kernel_ms"), "res", k.get("resolve_chain",{}).get("kernel_ms"), "serial", d.get("serial_step_ms",{}).get("total"))
except Exception as e: print("$tag FAILED", e)
PY
}
for rep in a b c; do
for c in 0 64 96 128; do run cus${c}_$rep PBSGPU_SCAN_CUS_SHARED=$c timeout 300 python bench.py --no-cpu-baseline --steps 24; done
done
