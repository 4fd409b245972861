#!/usr/bin/env python3
"""Top kernels of a rocprofv3 --kernel-trace --stats --output-format csv run: calls, mean duration, share of GPU time.

    python tools/kstats.py gpurun_out/prof [n]"""
import csv
import glob
import re
import sys

d = sys.argv[1]
n = int(sys.argv[2]) if len(sys.argv) > 2 else 25
f = glob.glob(d + "/**/*kernel_stats.csv", recursive=True)[0]
rows = list(csv.DictReader(open(f)))
tot = sum(int(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"]).split("(")[0][:64]
    print("%-66s %7d %9.1f us %5.1f%%" % (name, int(r["Calls"]), float(r["AverageNs"]) / 1e3, 100.0 * int(r["TotalDurationNs"]) / tot))
print("total %.3f ms" % (tot / 1e6))
