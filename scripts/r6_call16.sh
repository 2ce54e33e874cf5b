#!/bin/bash
# round 6, call 16: what does the loaded producer wait for? (a) address-translation counters of the service as an ordinary dispatch
# (every lane busy = the loaded state), (b) the probe's ns per block step with half / quarter of the lanes
out=gpurun_out/r6c16; mkdir -p $out
export PYTHONFAULTHANDLER=1
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
cd /tmp && export TMPDIR=/tmp
rocprofv3 --list-avail 2>/dev/null | grep -i -o "\b[A-Z_0-9]*\(UTCL\|TLB\|TRANSLATION\)[A-Za-z_0-9]*" | sort -u > $OUT/counters_tlb.txt; wc -l $OUT/counters_tlb.txt; head -40 $OUT/counters_tlb.txt | tr '\n' ' '
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
for set in "TCP_UTCL1_TRANSLATION_HIT_sum TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_REQUEST_sum" "TCP_TA_TCP_STATE_READ_sum TCP_PENDING_STALL_CYCLES_sum TCP_TCC_READ_REQ_LATENCY_sum TCP_TCC_READ_REQ_sum"; do
  tag=$(echo $set | cut -c1-24 | tr ' ' '_')
  timeout 300 rocprofv3 --pmc $set -d $OUT/pmc_$tag -o p -- python3 $ROOT/scripts/r4_ring_pmc.py 24 4 > $OUT/pmc_$tag.json 2> $OUT/pmc_$tag.err
  $EXP counters $(db $OUT/pmc_$tag) $OUT/pmc_$tag.csv 2>/dev/null
  grep -i "sha256" $OUT/pmc_$tag.csv | sed 's/void pbsk:://' | cut -c1-260
  tail -2 $OUT/pmc_$tag.err | cut -c1-200
done
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
cd $ROOT
line() { python3 - "$1" "$2" <<'PY'
import json,sys
for l in open(sys.argv[1]):
    if l.startswith('{'):
        d=json.loads(l); r=d['roofline']; rg=r.get('regime') or {}
        g=lambda ph,s,k='ns_per_block_step': ((rg.get(ph) or {}).get(s) or {}).get(k)
        print(sys.argv[2], d['value'], 'feed', r['feed_phase']['GiBps'], 'pair ns feed/drain/single', g('feed_phase','pair'), g('drain','pair'), g('single_file','pair'), 'sclk feed', g('feed_phase','pair','sclk_mhz'))
PY
}
for sha in 176 88 44; do
  PBSGPU_RING_XP_CUS=16 timeout 400 python bench.py --gpus 1 --steps 10 --warmup 3 --no-extras --no-cpu-baseline --ring-sha-cus $sha > $out/bench_sha$sha.json 2> $out/bench_sha$sha.err; line $out/bench_sha$sha.json "pair CUs $sha"
done
