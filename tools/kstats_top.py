#!/usr/bin/env python3
"""Top kernels of a rocprofv3 kernel_stats.csv: total ms, calls, average us (names shortened)."""
import csv
import re
import sys

rows = list(csv.DictReader(open(sys.argv[1])))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 30
tot = sum(float(r["TotalDurationNs"]) for r in rows)
for r in rows[:n]:
    name = re.sub(r"\(anonymous namespace\)::", "", r["Name"])
    name = re.sub(r"^void ", "", name)
    name = name.split("(")[0][:64]
    print("%-64s calls %6d  total %9.3f ms (%5.2f %%)  avg %8.1f us" % (name, int(r["Calls"]), float(r["TotalDurationNs"]) / 1e6,
                                                                      100 * float(r["TotalDurationNs"]) / tot, float(r["AverageNs"]) / 1e3))
