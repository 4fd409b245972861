#!/bin/bash
# Round-5 profile set (run on the GPU box from the repo root; one gpurun call = one box): the default bench line, rocprofv3 kernel stats
# and PMC FETCH_SIZE / WRITE_SIZE passes of the SAME command in its all-steps-in-step form (separate runs, kernel-trace only), the step
# timeline and decoder segment from that trace, the CTC op's timing / traffic attribution, op micro-benchmarks and the error tables
# DESIGN.md section 0 cites.
set -u
OUT=gpurun_out/r5prof
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
T="timeout -k 5"
$T 1200 python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
$T 300 python3 bench.py --steps 20 --warmup 3 --mode fwd --no-cpu-baseline --no-also > $OUT/bench_fwd.json 2>> $OUT/bench_train.err
$T 300 python3 bench.py --steps 5 --warmup 1 --mode decode --no-also > $OUT/bench_decode_s1.json 2>> $OUT/bench_train.err
$T 300 python3 tools/bench_ops.py > $OUT/bench_ops.jsonl 2>> $OUT/bench_train.err
$T 200 python3 tools/bench_ffn.py > $OUT/bench_ffn.txt 2>> $OUT/bench_train.err
$T 200 python3 tools/bench_vocab.py 2>&1 | grep -v amdgpu > $OUT/bench_ctc_branch.txt
$T 300 python3 tools/bench_wgrad_batch.py 2>&1 | grep -v amdgpu > $OUT/bench_wgrad_batch.txt
$T 300 python3 tools/attn_bwd_stats.py 2>&1 | grep -v amdgpu > $OUT/attn_bwd_stats.txt
$T 300 python3 tools/ctc_fullsize_err.py 2>&1 | grep -v amdgpu > $OUT/ctc_fullsize_err.txt
$T 300 python3 tools/g17_bf16_err.py 2>&1 | grep -v amdgpu > $OUT/g17_bf16_err.txt
$T 120 python3 tools/probe_bf16_split.py 2>&1 | grep -v amdgpu > $OUT/bf16_split.txt
( for spec in "X=0" "SIDE_INLINE=1"; do echo "== $spec"; ( export $spec; $T 200 python3 tools/step_segments.py 2>&1 | grep -v amdgpu ); done ) > $OUT/segments.txt 2>&1
( for v in 0 2 4 6; do echo "== ASR_AMD_CTC_DBG=$v"; ASR_AMD_CTC_DBG=$v $T 120 python3 tools/ab_ctc.py 2>&1 | grep -v amdgpu; done ) > $OUT/ctc_ab.txt 2>&1
bash tools/kt_ctc.sh default ASR_AMD_CTC_DBG=2 > $OUT/ctc_kernel_trace.txt 2>&1
cd /tmp
for v in 0 2; do
  for c in FETCH_SIZE WRITE_SIZE; do
    ( export ASR_AMD_CTC_DBG=$v; $T 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_${c}_$v -- python3 $R/tools/ab_ctc.py > /dev/null 2>&1 )
  done
done
$T 600 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also --per-op in_step > $R/$OUT/bench_train_profiled.json 2>/dev/null
$T 600 rocprofv3 --kernel-trace --pmc FETCH_SIZE --output-format csv -d $R/$OUT/pmc_fetch -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 --per-op in_step > /dev/null 2>&1
$T 600 rocprofv3 --kernel-trace --pmc WRITE_SIZE --output-format csv -d $R/$OUT/pmc_write -- python3 $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-also --graph 0 --per-op in_step > /dev/null 2>&1
$T 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt_attn_eval -- python3 $R/tools/prof_attn.py --bwd > /dev/null 2>&1
$T 300 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt_attn_train -- python3 $R/tools/prof_attn.py --bwd --drop > /dev/null 2>&1
$T 300 rocprofv3 --kernel-trace --stats --output-format csv -d $R/$OUT/kt_aishell -- python3 $R/bench.py --model aishell --brief --steps 20 --warmup 3 > /dev/null 2>&1
cd $R
python3 - > $OUT/ctc_pmc.txt 2>&1 <<PY
import csv, glob, collections
for v in (0, 2):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        vals = []
        for f in glob.glob("$OUT/pmc_%s_%d/**/*counter_collection.csv" % (c, v), recursive=True):
            for r in csv.DictReader(open(f)):
                if "ctc_fused" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    vals.append(float(r["Counter_Value"]))
        if vals:
            vals.sort()
            med = vals[len(vals) // 2]
            print("ASR_AMD_CTC_DBG=%d (%s) %-10s ctc_fused_fwd_kernel n=%d median %.1f MB (counter in KB; FETCH doubled per the gfx950 correction)" % (
                v, "labels gathered from the LDS image of the row" if v == 0 else "round 4: labels gathered by a second global load", c, len(vals),
                med * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6))
PY
KS=$(find $OUT/kt -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && cp $KS $OUT/bench_train_kernel_stats.csv && python3 tools/kstats_top.py $KS 40 > $OUT/bench_train_kernel_stats_top.txt
STEPS=$(python3 -c "import json,sys; print(json.loads([l for l in open('$OUT/bench_train_profiled.json') if l.startswith('{')][-1])['steps_executed'])")
python3 tools/roofline_from_csv.py $OUT/bench_train_profiled.json $OUT/bench_train_kernel_stats.csv --csv-steps $STEPS > $OUT/roofline_from_csv.txt 2>&1
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1); [ -n "$KT" ] && python3 tools/timeline.py $KT adam_dev | cut -c1-160 > $OUT/step_timeline.txt 2>&1
python3 tools/decoder_segment.py $OUT/kt --list > $OUT/decoder_segment.txt 2>&1
F=$(find $OUT/pmc_fetch -name "*counter_collection.csv" | head -1); W=$(find $OUT/pmc_write -name "*counter_collection.csv" | head -1)
[ -n "$F" ] && [ -n "$W" ] && python3 tools/pmc_summary.py $F $W $OUT/pmc_traffic_train_s1.json > $OUT/pmc_summary.txt 2>&1
KS=$(find $OUT/kt_aishell -name "*kernel_stats.csv" | head -1); [ -n "$KS" ] && python3 tools/kstats_top.py $KS 24 > $OUT/aishell_kernel_stats_top.txt
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
rm -rf $OUT/kt $OUT/pmc_fetch $OUT/pmc_write $OUT/pmc_FETCH_SIZE_* $OUT/pmc_WRITE_SIZE_* $OUT/kt_attn_eval $OUT/kt_attn_train $OUT/kt_aishell gpurun_out/kt_ctc
ls -la $OUT
python3 - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_train.json") if l.startswith("{")][-1])
print("S1 ms/step", d["ms_per_step"], "value", d["value"])
print("roofline", {k: d["roofline"].get(k) for k in ("kernel", "frac", "frac_in_step", "ms_per_step", "ms_per_step_in_step")})
print("ctc", json.dumps(d["ctc"]["branch_ms_per_call"]), d["ctc"]["fwd_ms_standalone"], d["ctc"]["fwd_frac_of_hbm_peak_standalone"], d["ctc"]["fwd_ms_standalone_after_run"])
print("also", {k: v.get("ms_per_step") for k, v in (d.get("also") or {}).items()})
print("cpu", d.get("cpu_baseline", {}).get("value"), d.get("parity_vs_oracle_max_abs"))
PY
cat $OUT/roofline_from_csv.txt | tail -3
