#!/bin/bash
# kernel totals (rocprofv3 --kernel-trace --stats) of the other workloads' training steps: --model s2 / cif / aishell
set -u
OUT=gpurun_out/models
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
T="timeout -k 5"
for m in "$@"; do
  cd /tmp; $T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_$m -- python3 $R/bench.py --model $m --brief --steps 20 --warmup 3 > $R/$OUT/bench_$m.json 2>/dev/null; cd $R
  KS=$(find $OUT/kt_$m -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && python3 tools/kstats_top.py $KS 28 > $OUT/${m}_kernel_stats_top.txt; rm -rf $OUT/kt_$m
  python3 -c "import json; d=json.loads([l for l in open('$OUT/bench_$m.json') if l.startswith('{')][-1]); print('$m', d['ms_per_step'], d['config']['launch'][:60])"
  cat $OUT/${m}_kernel_stats_top.txt
done
