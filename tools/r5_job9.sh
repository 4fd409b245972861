#!/bin/bash
set -u
OUT=gpurun_out/r5job9
mkdir -p $OUT
export TMPDIR=/tmp
T="timeout -k 5"
R=$GRAFT_REPO_ROOT
cd /tmp; $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_aishell -- python3 $R/bench.py --model aishell --brief --steps 20 --warmup 3 > $R/$OUT/aishell_bench.json 2>/dev/null; cd $R
KS=$(find $OUT/kt_aishell -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && python3 tools/kstats_top.py $KS 32 > $OUT/aishell_kernel_stats.txt; rm -rf $OUT/kt_aishell
tail -c 400 $OUT/aishell_bench.json; echo; cat $OUT/aishell_kernel_stats.txt
