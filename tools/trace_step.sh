#!/bin/bash
# One replayed training step's kernel timeline (gaps, shares): rocprofv3 kernel trace of the default bench, cut by tools/timeline.py.
set -u
OUT=gpurun_out/${TRACE_OUT:-trace_step}
mkdir -p $OUT
export TMPDIR=/tmp
R=$PWD
cd /tmp
timeout -k 5 600 rocprofv3 --kernel-trace --output-format csv -d $R/$OUT/kt -- python3 $R/bench.py --steps 20 --warmup 3 --no-cpu-baseline --no-also ${BENCH_ARGS:-} > $R/$OUT/bench.json 2> $R/$OUT/bench.err
cd $R
KT=$(find $OUT/kt -name "*kernel_trace.csv" | head -1)
[ -n "$KT" ] && python3 tools/timeline.py $KT adam_dev > $OUT/step_timeline.txt 2>&1
[ -n "$KT" ] && python3 - $KT > $OUT/step_kernels.txt <<'PY'
import csv, sys, re
rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Stream_Id", "0")) for r in csv.DictReader(open(sys.argv[1]))]
rows.sort()
cuts = [i for i, r in enumerate(rows) if "adam_dev" in r[2]]
lo, hi = cuts[-3] + 1, cuts[-2] + 1
t0 = rows[lo][0]
for s, e, n, st in rows[lo:hi]:
    n = re.sub(r"\(anonymous namespace\)::", "", n); n = re.sub(r"^void ", "", n)
    print("%9.1f %8.1f s%-3s %s" % ((s - t0) / 1e3, (e - s) / 1e3, st, n.split("(")[0][:70]))
PY
rm -rf $OUT/kt
tail -3 $OUT/bench.json
