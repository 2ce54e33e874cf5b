#!/bin/bash
# Round-4 profile evidence (through gpurun):  scripts/r4_profile.sh OUTDIR
#   1. rocprofv3 --kernel-trace --stats of the driver's command (ring, fused control kernel)
#   2. PMC passes on the ring's OWN kernels (scripts/r4_ring_pmc.py: the service as an ordinary dispatch):
#      SQ counters, FETCH_SIZE, WRITE_SIZE in separate runs — closes "roofline.traffic of the ring is derived, not measured"
ROOT=${GRAFT_REPO_ROOT:-$PWD}
OUT=$ROOT/${1:-gpurun_out/r4prof}
mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
EXP="python3 $ROOT/scripts/rocpd_export.py"
db() { find $1 -name "*_results.db" | head -1; }
timeout 500 rocprofv3 --kernel-trace --stats -d $OUT/k_default -o bench -- python3 $ROOT/bench.py --gpus 1 --steps 20 --warmup 5 --no-extras > $OUT/bench_default_traced.json 2> $OUT/bench_default_traced.err
$EXP stats $(db $OUT/k_default) $OUT/kernel_stats_bench_default.csv; $EXP trace $(db $OUT/k_default) $OUT/kernel_trace_bench_default.csv
gzip -f $OUT/kernel_trace_bench_default.csv
timeout 300 python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_plain.json 2> $OUT/ring_pmc_plain.err; cat $OUT/ring_pmc_plain.json
timeout 400 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_WAVES SQ_WAVE_CYCLES SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_BUSY_CYCLES -d $OUT/pmc_sq -o s -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_sq.json 2> $OUT/ring_pmc_sq.err
$EXP counters $(db $OUT/pmc_sq) $OUT/pmc_sq_ring.csv
timeout 400 rocprofv3 --pmc FETCH_SIZE -d $OUT/pmc_fetch -o f -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_fetch.json 2> $OUT/ring_pmc_fetch.err
$EXP counters $(db $OUT/pmc_fetch) $OUT/pmc_fetch_size_ring.csv
timeout 400 rocprofv3 --pmc WRITE_SIZE -d $OUT/pmc_write -o w -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_write.json 2> $OUT/ring_pmc_write.err
$EXP counters $(db $OUT/pmc_write) $OUT/pmc_write_size_ring.csv
timeout 400 rocprofv3 --pmc GRBM_GUI_ACTIVE GRBM_COUNT -d $OUT/pmc_grbm -o g -- python3 $ROOT/scripts/r4_ring_pmc.py 48 4 > $OUT/ring_pmc_grbm.json 2> $OUT/ring_pmc_grbm.err
$EXP counters $(db $OUT/pmc_grbm) $OUT/pmc_grbm_ring.csv
find $OUT -name "*.db" -delete; find $OUT -type d -empty -delete
head -8 $OUT/kernel_stats_bench_default.csv | cut -c1-200
for f in sq fetch_size write_size grbm; do grep -i "sha256_pair\|scan3\|ring_fill\|ring_control" $OUT/pmc_${f}_ring.csv | cut -c1-220; done
cat $OUT/ring_pmc_sq.json $OUT/ring_pmc_fetch.json | cut -c1-400
du -sh $OUT
