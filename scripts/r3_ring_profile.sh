#!/bin/bash
# rocprofv3 kernel trace of a short ring bench (run on the GPU box): per-kernel time of the cut rounds next to the service
out=${1:-gpurun_out/r3prof}; shift
mkdir -p $out
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
timeout 600 rocprofv3 --kernel-trace --stats -d $R/$out/trace -o ring -- python $R/bench.py --steps 8 --warmup 2 --no-extras --no-cpu-baseline "$@" > $R/$out/bench.json 2> $R/$out/bench.err
cd $R
f=$(find $out/trace -name "*kernel_stats.csv" | head -1)
[ -n "$f" ] && cp $f $out/kernel_stats.csv && head -25 $f | cut -c1-200
tail -c 400 $out/bench.json
