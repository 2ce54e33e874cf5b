timeout 600 python bench.py --workload verify --file-mib 64 --gib 32 --steps 2 2>/dev/null | python -c "
import json,sys; d=json.load(sys.stdin); r=d['results']; print({k:r[k] for k in r if k not in ('note',)})"
