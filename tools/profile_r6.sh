#!/bin/bash
# Round-6 profile set (run on the GPU box from the repo root; one gpurun call = one box).  "quick": only the kernel trace of the bench in its
# all-steps-in-step form + the step timeline and the decoder segment from it.
set -u
OUT=gpurun_out/r6prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
T="timeout -k 5"
MODE=${1:-full}
if [ "$MODE" = full ]; then
  $T 1200 python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
  $T 300 python3 tools/check_ffn2.py 2>&1 | grep -v amdgpu > $OUT/ffn2_check.txt
  $T 300 python3 tools/bench_ops.py > $OUT/bench_ops.jsonl 2>> $OUT/bench_train.err
  ( for spec in "X=0" "SIDE_INLINE=1"; do echo "== $spec"; ( export $spec; $T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu ); done ) > $OUT/segments.txt 2>&1
  ( $T 120 python3 tools/ab_ctc.py 2>&1 | grep -v amdgpu ) > $OUT/ctc_ab.txt 2>&1
  bash tools/kt_ctc.sh default > $OUT/ctc_kernel_trace.txt 2>&1
fi
cd /tmp
$T 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --per-op in_step > $R/$OUT/bench_train_profiled.json 2>/dev/null
if [ "$MODE" = full ]; then
  $T 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 --per-op in_step > /dev/null 2>&1
  $T 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 --per-op in_step > /dev/null 2>&1
  # the feed-forward forward's matrix-pipe share: SQ_VALU_MFMA_BUSY_CYCLES (cycles) against SQ_BUSY_CU_CYCLES / the kernel's duration
  $T 300 rocprofv3 --kernel-trace --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CU_CYCLES SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY --output-format csv -d $R/$OUT/pmc_ffn -- python3 $R/tools/check_ffn2.py --time-only > /dev/null 2>&1
fi
cd $R
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $OUT/bench_train_kernel_stats.csv && python3 tools/kstats_top.py $KS 40 > $OUT/bench_train_kernel_stats_top.txt
STEPS=$(python3 -c "import json,sys; print(json.loads([l for l in open('$OUT/bench_train_profiled.json') if l.startswith('{')][-1])['steps_executed'])")
python3 tools/roofline_from_csv.py $OUT/bench_train_profiled.json $OUT/bench_train_kernel_stats.csv --csv-steps $STEPS > $OUT/roofline_from_csv.txt 2>&1
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python3 tools/timeline.py $KT adam_dev | cut -c1-160 > $OUT/step_timeline.txt 2>&1
python3 tools/decoder_segment.py $OUT/kt --list > $OUT/decoder_segment.txt 2>&1
if [ "$MODE" = full ]; then
  F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
  [ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic_train_s1.json > $OUT/pmc_summary.txt 2>&1
  python3 - > $OUT/ffn2_pmc.txt 2>&1 <<PY
import csv, glob, collections
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob("$OUT/pmc_ffn/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "ffn_fwd2" in r["Kernel_Name"]:
            acc[(r["Kernel_Name"][:60], r.get("Grid_Size", ""))][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, d in acc.items():
    print(k)
    for c, v in sorted(d.items()):
        v = sorted(v)
        print("   %-28s n=%d median %.4g" % (c, len(v), v[len(v) // 2]))
PY
fi
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_ffn gpurun_out/kt_ctc
ls -la $OUT
