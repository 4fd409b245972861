"""Which kernels of a training step run on which (hardware queue, stream) - from a rocprofv3 --kernel-trace CSV; the light streams in full.

  python tools/step_streams.py <kernel_trace.csv>"""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:56]


rows = []
for r in csv.DictReader(open(sys.argv[1])):
    rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
rows.sort()
cuts = [i for i, r in enumerate(rows) if "adam_dev" in r[2]]
c0, c1 = cuts[len(cuts) // 2], cuts[len(cuts) // 2 + 1]
step = rows[c0 + 1:c1 + 1]
t0 = rows[c0][1]
by = defaultdict(list)
for r in step:
    by[(r[3], r[4])].append(r)
for k, v in sorted(by.items(), key=lambda kv: -len(kv[1])):
    print("queue %s stream %s: %d kernels, busy %.3f ms" % (k[0], k[1], len(v), sum(r[1] - r[0] for r in v) / 1e6))
    if len(v) <= 40:
        for s, e, n, q, st in v:
            print("    start %9.1f us  dur %7.1f  %s" % ((s - t0) / 1e3, (e - s) / 1e3, short(n)))
