#!/bin/bash
# kernel-trace durations (us) of the attention kernels of tools/prof_attn.py --bwd [--drop]
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
rm -rf $R/gpurun_out/kt_attn
rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_attn -- python3 $R/tools/prof_attn.py --bwd "$@" > /dev/null 2>&1
python3 - <<PY
import csv, glob, collections, re
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/kt_attn/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"attn_\w+(<[^>]*>)?", n)
        if m:
            acc[m.group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print("%-62s n=%d  median %.1f us  min %.1f" % (k, len(v), v[len(v) // 2], v[0]))
PY
