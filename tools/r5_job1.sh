#!/bin/bash
# round 5, GPU job 1: new parity tests, CTC gather A/B (+ FETCH_SIZE attribution), bf16 split probe, the default bench line
set -u
OUT=gpurun_out/r5job1
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
T="timeout -k 5"
$T 900 python3 -m pytest tests -m gpu -q > $OUT/pytest.log 2>&1; echo "pytest rc $?" >> $OUT/pytest.log
tail -5 $OUT/pytest.log
$T 120 python3 tools/probe_bf16_split.py > $OUT/bf16_split.txt 2>&1
$T 300 python3 tools/attn_bwd_stats.py > $OUT/attn_bwd_stats.txt 2>&1
for v in 0 2 4 6; do echo "== ASR_AMD_CTC_DBG=$v"; ASR_AMD_CTC_DBG=$v $T 120 python3 tools/ab_ctc.py; done > $OUT/ctc_ab.txt 2>&1
for hs in 0 2 4 6 8 12 16 24; do echo "== ASR_AMD_CTC_HEADSTART=$hs"; ASR_AMD_CTC_HEADSTART=$hs $T 120 python3 tools/ab_ctc.py; done >> $OUT/ctc_ab.txt 2>&1
( bash tools/kt_ctc.sh default ASR_AMD_CTC_DBG=2 ASR_AMD_CTC_HEADSTART=6 ASR_AMD_CTC_HEADSTART=12 ) > $OUT/ctc_kernel_trace.txt 2>&1
cd /tmp
for v in 0 2 8; do
  for c in FETCH_SIZE WRITE_SIZE; do
    rm -rf $R/$OUT/pmc_${c}_$v
    ( export ASR_AMD_CTC_DBG=$v; $T 300 rocprofv3 --kernel-trace --pmc $c --output-format csv -d $R/$OUT/pmc_${c}_$v -- python3 $R/tools/ab_ctc.py > /dev/null 2>&1 )
  done
done
cd $R
python3 - > $OUT/ctc_pmc.txt 2>&1 <<PY
import csv, glob, collections
for v in (0, 2, 8):
    for c in ("FETCH_SIZE", "WRITE_SIZE"):
        acc = collections.defaultdict(list)
        for f in glob.glob("$OUT/pmc_%s_%d/**/*counter_collection.csv" % (c, v), recursive=True):
            for r in csv.DictReader(open(f)):
                if "ctc_fused" in r["Kernel_Name"] and r["Counter_Name"] == c:
                    acc[r["Kernel_Name"][:40]].append(float(r["Counter_Value"]))
        for k, vals in acc.items():
            vals = sorted(vals)
            # counter unit: FETCH_SIZE / WRITE_SIZE in KB... printed raw + x64 B / x32 B readings; MI355X guide: see tools/pmc_summary.py for the unit used in profiles/
            med = vals[len(vals) // 2]; print("dbg=%d %-10s %-40s n=%d median %.1f MB (counter in KB; FETCH doubled per the gfx950 correction)" % (v, c, k, len(vals), med * 1024 * (2 if c == "FETCH_SIZE" else 1) / 1e6))
PY
rm -rf $OUT/pmc_*
$T 1200 python3 bench.py > $OUT/bench_train.json 2> $OUT/bench_train.err
tail -c 600 $OUT/bench_train.err
python3 - <<PY
import json
d = json.loads([l for l in open("$OUT/bench_train.json") if l.startswith("{")][-1])
print("S1 ms/step", d["ms_per_step"], "value", d["value"], "steps_executed", d.get("steps_executed"))
print("roofline", {k: d["roofline"][k] for k in ("kernel", "frac", "frac_in_step", "ms_per_step", "ms_per_step_in_step")})
print("ctc", json.dumps(d["ctc"]["branch_ms_per_call"]), d["ctc"]["fwd_ms_standalone"], d["ctc"]["fwd_frac_of_hbm_peak_standalone"])
print("also", {k: v.get("ms_per_step") for k, v in (d.get("also") or {}).items()})
for k in d["kernels"]: print(k["name"], k["ms_per_call"], k["ms_per_step"])
PY
