#!/bin/bash
# round 4, GPU call 4: new round-4 tests, host-feed archive transitions traced, progressive page release on configs[2], configs[4]
# through the ring, near-steady-state SQ counters of the service (48 service CUs: ~9 chunks per lane)
out=gpurun_out/r4c4; mkdir -p $out
export PYTHONFAULTHANDLER=1
( time timeout 600 python -m pytest tests/test_gpu_round4.py tests/test_gpu_ring.py -m gpu -x -q --timeout 300 ) > $out/pytest.log 2>&1; tail -15 $out/pytest.log | cut -c1-250
PBS_BENCH_HF_TRACE=1 timeout 200 python bench.py --workload hostfeed --producers 1 --archives 4 --steps 96 --warmup 4 > $out/hf1x4.json 2> $out/hf1x4.err; grep "hostfeed trace" $out/hf1x4.err; python3 -c "
import json
d=json.loads([l for l in open('$out/hf1x4.json') if l.startswith('{')][0]); print('hf 1x4', d['value'], d['roofline']['frac_of_measured_h2d'], d['write_phase'])"
PBS_BENCH_HF_TRACE=1 timeout 200 python bench.py --workload hostfeed --producers 1 --archives 1 --steps 96 --warmup 4 > $out/hf1.json 2> $out/hf1.err; grep "hostfeed trace" $out/hf1.err; python3 -c "
import json
d=json.loads([l for l in open('$out/hf1.json') if l.startswith('{')][0]); print('hf 1x1', d['value'], d['roofline']['frac_of_measured_h2d'], d['write_phase'])"
for wl in ring_manyfiles ring_rechunk ring_corpus_dup; do
  timeout 300 python bench.py --workload $wl --steps 6 --warmup 1 > $out/$wl.json 2> $out/$wl.err; tail -2 $out/$wl.err
  python3 -c "
import json
d=json.loads([l for l in open('$out/$wl.json') if l.startswith('{')][0]); print('$wl', d['value'], d['roofline']['feed_phase'], d.get('results'), d.get('cpu_baseline',{}).get('records_match_gpu'), d.get('cpu_baseline',{}).get('records_checked'))"
done
timeout 300 python bench.py --gpus 1 --steps 20 --warmup 5 --no-extras --no-cpu-baseline > $out/bench_default.json 2>/dev/null; python3 -c "
import json
d=json.loads([l for l in open('$out/bench_default.json') if l.startswith('{')][0]); r=d['roofline']; print('default', d['value'], r['feed_phase']['GiBps'], r['feed_phase']['drain_seconds'], r['single_file']['ms'])"
cd /tmp && export TMPDIR=/tmp
ROOT=$GRAFT_REPO_ROOT; OUT=$ROOT/$out
PBSGPU_RING_SHA_CUS=48 timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq48 -o s -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_sq48.json 2> $OUT/ring_pmc_sq48.err
python3 $ROOT/scripts/rocpd_export.py counters $(find $OUT/pmc_sq48 -name "*_results.db" | head -1) $OUT/pmc_sq_ring_48cus.csv
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
grep sha256_pair $OUT/pmc_sq_ring_48cus.csv | cut -c150-260; cat $OUT/ring_pmc_sq48.json
