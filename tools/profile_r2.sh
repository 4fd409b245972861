#!/bin/bash
# Round-2 profile set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats of the same command, PMC
# FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace domains besides kernel-trace), attention counters.
set -u
OUT=gpurun_out/r2prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
python3 bench.py --steps 20 --warmup 3 > $OUT/bench_train.json 2> $OUT/bench_train.err
python3 bench.py --steps 20 --warmup 3 --mode fwd --no-cpu-baseline > $OUT/bench_fwd.json 2>> $OUT/bench_train.err
python3 bench.py --steps 10 --warmup 2 --model s2 --no-cpu-baseline > $OUT/bench_train_s2.json 2>> $OUT/bench_train.err
python3 bench.py --steps 5 --warmup 1 --mode decode --model s2 > $OUT/bench_decode_s2.json 2>> $OUT/bench_train.err
python3 bench.py --steps 5 --warmup 1 --mode decode > $OUT/bench_decode_s1.json 2>> $OUT/bench_train.err
python3 bench.py --steps 5 --warmup 1 --mode decode --beam 5 > $OUT/bench_decode_s1_beam5.json 2>> $OUT/bench_train.err
python3 bench.py --steps 5 --warmup 1 --mode decode --model cif > $OUT/bench_decode_cif.json 2>> $OUT/bench_train.err
python3 bench.py --steps 10 --warmup 2 --model cif > $OUT/bench_train_cif.json 2>> $OUT/bench_train.err
python3 tools/bench_ops.py > $OUT/bench_ops.jsonl 2>> $OUT/bench_train.err
cd /tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline > /dev/null 2>&1
rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES --output-format csv -d $R/$OUT/pmc_attn -- python3 $R/tools/prof_attn.py > /dev/null 2>&1
rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/ktd -- python3 $R/bench.py --steps 5 --warmup 1 --mode decode --no-cpu-baseline > /dev/null 2>&1
cd $R
KD=$(find $OUT/ktd -name "*kernel_stats.csv" | head -1); [ -n "$KD" ] && cp $KD $OUT/bench_decode_s1_kernel_stats.csv
rm -rf $OUT/ktd
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $OUT/bench_train_kernel_stats.csv
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic_train_s1.json > $OUT/pmc_summary.txt 2>&1
A=$(find $OUT/pmc_attn -name "*counter_collection.csv" | head -1); [ -n "$A" ] && cp $A $OUT/attn_counters.csv
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_attn
ls -la $OUT
