"""The end of a training step from a rocprofv3 --kernel-trace CSV: the last N kernels before the step's fused Adam, with queue, start and end
relative to the Adam launch - how long the weight-gradient stream runs on after the main chain's last kernel.

  python tools/step_tail.py <kernel_trace.csv> [N=30]"""
import csv
import re
import sys


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    return re.sub(r"^void ", "", n).split("(")[0][:56]


def main():
    path, N = sys.argv[1], int(sys.argv[2]) if len(sys.argv) > 2 else 30
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if "adam_dev" in r[2]]
    if len(cuts) < 3:
        print("not enough steps"); return
    tails = []
    for c in cuts[2:-1]:
        t_adam = rows[c][0]
        seg = rows[max(0, c - 400):c]
        mainq = max(set(r[3] for r in seg), key=lambda q: sum(1 for r in seg if r[3] == q))
        last_main = max((r[1] for r in seg if r[3] == mainq), default=t_adam)
        tails.append((t_adam - last_main) / 1e3)
    tails.sort()
    print("main chain's last kernel end -> Adam start over %d steps: median %.1f us, min %.1f, max %.1f" % (len(tails), tails[len(tails) // 2], tails[0], tails[-1]))
    c = cuts[len(cuts) // 2]
    t_adam = rows[c][0]
    for s, e, n, q, st in rows[c - N:c + 1]:
        print("  q%s  start %9.1f us  end %9.1f us  dur %7.1f  %s" % (q, (s - t_adam) / 1e3, (e - t_adam) / 1e3, (e - s) / 1e3, short(n)))
    print("the step's head (times from the end of the previous step's Adam):")
    t_end = rows[c][1]
    for s, e, n, q, st in rows[c + 1:c + 1 + N]:
        print("  q%s  start %9.1f us  end %9.1f us  dur %7.1f  %s" % (q, (s - t_end) / 1e3, (e - t_end) / 1e3, (e - s) / 1e3, short(n)))


if __name__ == "__main__":
    main()
