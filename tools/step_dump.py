#!/usr/bin/env python3
"""Chronological kernel list of ONE training step from a rocprofv3 --kernel-trace CSV (adam marks the step's end):
start offset, duration, gap to the previous kernel's end on the same queue, queue, name.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 6 --warmup 2 --graph 0
    python tools/step_dump.py gpurun_out/tl [step_index] > gpurun_out/step_dump.txt
"""
import csv
import glob
import sys


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    head = n.split("(")[0]
    return head[:70]


def main(d, k=6):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"],
                     r.get("Grid_Size", ""), r.get("Workgroup_Size", "")))
    rows.sort()
    ends = [e for s, e, q, n, g, w in rows if "adam" in n[:60]]
    t0, t1 = ends[k], ends[k + 1]
    last = {}
    qn = {}
    print(f"# step {k}: wall {(t1 - t0) / 1e6:.3f} ms")
    for s, e, q, n, g, w in rows:
        if s < t0 or e > t1 + 1:
            continue
        qi = qn.setdefault(q, len(qn))
        gap = (s - last[q]) / 1e3 if q in last else 0.0
        last[q] = e
        print(f"{(s - t0) / 1e3:9.1f} {(e - s) / 1e3:8.1f} {gap:7.1f} q{qi} {g:>8}/{w:<4} {short(n)}")


if __name__ == "__main__":
    main(sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 6)
