#!/bin/bash
# kernel-trace durations (us) of the CTC forward kernels of tools/ab_ctc.py, one line per environment given as "VAR=val,VAR=val" arguments
cd /tmp && export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
for spec in "${@:-default}"; do
  rm -rf $R/gpurun_out/kt_ctc
  ( if [ "$spec" != default ]; then export ${spec//,/ }; fi
    rocprofv3 --kernel-trace --output-format csv -d $R/gpurun_out/kt_ctc -- python3 $R/tools/ab_ctc.py > /dev/null 2>&1 )
  python3 - "$spec" <<PY
import csv, glob, collections, re, sys
acc = collections.defaultdict(list)
for f in glob.glob("$R/gpurun_out/kt_ctc/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        n = r["Kernel_Name"]
        m = re.search(r"ctc_\w+(<[^>]*>)?", n)
        if m:
            acc[m.group(0)].append((int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) / 1e3)
for k, v in acc.items():
    v = sorted(v)
    print("%-28s %-40s n=%d  median %.1f us  min %.1f  p90 %.1f" % (sys.argv[1], k, len(v), v[len(v) // 2], v[0], v[int(len(v) * 0.9)]))
PY
done
