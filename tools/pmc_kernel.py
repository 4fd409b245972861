"""Summarise rocprofv3 --pmc counter_collection csv: mean per-dispatch counter values per kernel.  usage: pmc_kernel.py DIR [substr]"""
import csv, glob, sys, collections
d = sys.argv[1]; sub = sys.argv[2] if len(sys.argv) > 2 else ""
acc = collections.defaultdict(lambda: collections.defaultdict(list))
for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if sub in r["Kernel_Name"]:
            acc[r["Kernel_Name"][:int(sys.argv[3]) if len(sys.argv) > 3 else 60]][r["Counter_Name"]].append(float(r["Counter_Value"]))
for k, cs in acc.items():
    print(k)
    for c, v in sorted(cs.items()):
        v = v[1:] if len(v) > 2 else v
        print("   %-32s %16.0f  (n=%d)" % (c, sum(v) / len(v), len(v)))
