"""Mean FETCH_SIZE per launch of the slab weight-gradient kernels by grid size, from a rocprofv3 --pmc FETCH_SIZE counter_collection.csv
(KB units, doubled per the gfx950 correction of MI355X_MICROARCH.md)."""
import collections
import csv
import sys

acc = collections.defaultdict(list)
for r in csv.DictReader(open(sys.argv[1])):
    if "gemm_tn_v2" in r["Kernel_Name"] and r["Counter_Name"] == "FETCH_SIZE":
        acc[(r["Kernel_Name"].split("(")[0][-28:], int(r["Grid_Size"]))].append(float(r["Counter_Value"]) * 2 * 1024 / 1e6)
for k, v in sorted(acc.items()):
    print("%-30s grid %7d  n %4d  mean fetch MB %.1f" % (k[0], k[1], len(v), sum(v) / len(v)))
