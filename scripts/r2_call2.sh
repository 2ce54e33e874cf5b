set -x
mkdir -p gpurun_out/r2c2
run() { tag=$1; shift; timeout 600 "$@" > gpurun_out/r2c2/$tag.json 2> gpurun_out/r2c2/$tag.err; echo "rc=$?"; tail -c 600 gpurun_out/r2c2/$tag.err; python - <<PY
import json
try:
    d=json.load(open("gpurun_out/r2c2/$tag.json")); r=d["roofline"]; print("$tag", d["value"], d["ms_per_step"], r["frac"], r.get("latency_bound",{}).get("bound_GiBps"), d.get("serial_value"), d.get("results"), d.get("cpu_baseline",{}).get("records_match_gpu"), d.get("cpu_baseline",{}).get("value"))
except Exception as e: print("$tag FAILED", e)
PY
}
run s_stream python bench.py --gib 1 --slots 2 --steps 4 --warmup 1 --cpu-sample-gib 0.25
run s_many python bench.py --workload manyfiles --gib 1 --file-mib 8 --steps 4 --warmup 1 --cpu-sample-gib 0.25
run s_dup python bench.py --workload corpus_dup --gib 1 --file-mib 8 --steps 4 --warmup 2 --cpu-sample-gib 0.25
run s_rechunk python bench.py --workload rechunk --gib 1 --file-mib 8 --steps 4 --warmup 1 --cpu-sample-gib 0.25
run f_stream python bench.py
run f_stream_fifo python bench.py --collect fifo --no-cpu-baseline
run f_many python bench.py --workload manyfiles
run f_dup python bench.py --workload corpus_dup
run f_rechunk python bench.py --workload rechunk
run f_reread16 python bench.py --reread 16 --steps 32 --no-cpu-baseline
