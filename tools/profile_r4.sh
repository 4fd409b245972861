#!/bin/bash
# Round-4 profile set (run on the GPU box from the repo root): bench lines, rocprofv3 kernel stats of the same command, PMC
# FETCH_SIZE / WRITE_SIZE passes (separate runs, no trace domains besides kernel-trace), a step timeline, op micro-benchmarks.
set -u
OUT=gpurun_out/r4prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
T="timeout -k 5"
$T 900 python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
$T 300 python3 bench.py --steps 20 --warmup 3 --mode fwd --no-cpu-baseline --no-also > $OUT/bench_fwd.json 2>> $OUT/bench_train.err
$T 300 python3 bench.py --steps 5 --warmup 1 --mode decode --no-also > $OUT/bench_decode_s1.json 2>> $OUT/bench_train.err
$T 300 python3 bench.py --steps 5 --warmup 1 --mode decode --beam 5 --no-also > $OUT/bench_decode_s1_beam5.json 2>> $OUT/bench_train.err
$T 600 python3 tools/bench_ops.py > $OUT/bench_ops.jsonl 2>> $OUT/bench_train.err
$T 300 python3 tools/bench_ffn.py > $OUT/bench_ffn.txt 2>> $OUT/bench_train.err
$T 120 python3 tools/bench_heads.py > $OUT/bench_heads.txt 2>> $OUT/bench_train.err
$T 120 python3 tools/bench_dgrad_rows.py > $OUT/bench_dgrad_rows.txt 2>> $OUT/bench_train.err
$T 120 python3 tools/ctc_timeline.py > $OUT/ctc_timeline.txt 2>> $OUT/bench_train.err
$T 300 python3 tools/rccl_sanity.py --exec 2>&1 | grep -v "^RCCL\|^HIP\|^ROCm\|^Host\|^Libr\|amdgpu.ids\|socket.cpp" > $OUT/rccl_executor_world1.txt
$T 600 python3 tools/dp_hang_hunt.py --runs 16 --limit 90 > $OUT/dp_hang_hunt.txt 2>&1
bash tools/kt_ctc.sh default > $OUT/ctc_kernel_trace.txt 2>&1
cd /tmp
$T 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also > /dev/null 2>&1
$T 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 > /dev/null 2>&1
$T 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 > /dev/null 2>&1
$T 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_VALU_MFMA_MOPS_BF16 SQ_INSTS_VALU SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAVE_CYCLES SQ_VALU_MFMA_BUSY_CYCLES SQ_WAIT_ANY SQ_ACTIVE_INST_VALU --output-format csv -d $R/$OUT/pmc_attn -- python3 $R/tools/prof_attn.py --bwd --drop > /dev/null 2>&1
$T 600 rocprofv3 --kernel-trace --pmc SQ_INSTS_LDS SQ_INSTS_SALU SQ_ACTIVE_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_MFMA SQ_INSTS_VMEM GRBM_GUI_ACTIVE --output-format csv -d $R/$OUT/pmc_attn2 -- python3 $R/tools/prof_attn.py --bwd --drop > /dev/null 2>&1
$T 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt_attn_eval -- python3 $R/tools/prof_attn.py --bwd > /dev/null 2>&1
$T 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt_attn_train -- python3 $R/tools/prof_attn.py --bwd --drop > /dev/null 2>&1
cd $R
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $OUT/bench_train_kernel_stats.csv
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python3 tools/timeline.py $KT adam_dev | cut -c1-160 > $OUT/step_timeline.txt 2>&1
python3 tools/decoder_segment.py $OUT/kt --list > $OUT/decoder_segment.txt 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic_train_s1.json > $OUT/pmc_summary.txt 2>&1
( python3 tools/pmc_kernel.py $OUT/pmc_attn attn 48; python3 tools/pmc_kernel.py $OUT/pmc_attn2 attn 48 ) > $OUT/attn_counters.txt 2>&1
python3 - > $OUT/attn_kernel_times.txt <<PY
import csv, glob, collections, re
for mode in ("eval", "train"):
    acc = collections.defaultdict(list)
    for f in glob.glob("$OUT/kt_attn_%s/**/*kernel_trace.csv" % mode, recursive=True):
        for r in csv.DictReader(open(f)):
            m = re.search(r"attn_\\w+(<[^>]*>)?", r["Kernel_Name"])
            if m:
                acc[m.group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
    for k, v in sorted(acc.items()):
        v = sorted(v)
        print("%-6s %-44s n=%d  median %.1f us  min %.1f us" % (mode, k, len(v), v[len(v) // 2], v[0]))
PY
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_attn $OUT/pmc_attn2 $OUT/kt_attn_eval $OUT/kt_attn_train
ls -la $OUT
