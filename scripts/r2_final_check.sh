timeout 1800 python -X faulthandler -m pytest tests -x -q -m gpu 2>&1 | tail -3
for rep in a b; do for v in fused unfused; do
if [ $v = unfused ]; then export PBSGPU_COMPACT_UNFUSED=1; else unset PBSGPU_COMPACT_UNFUSED; fi
timeout 300 python bench.py --steps 24 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; print('$v', d['value'], d['ms_per_step'], 'res', k['resolve_chain'], 'serial', d['serial_step_ms'])"
done; done
unset PBSGPU_COMPACT_UNFUSED
timeout 300 python bench.py --workload manyfiles --no-cpu-baseline 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['roofline']; k=r['kernels']; print('manyfiles', d['value'], 'res', k['resolve_chain'])"
