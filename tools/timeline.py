"""Step timeline from a rocprofv3 --kernel-trace CSV: where the wall time of one training step goes.

  python tools/timeline.py <kernel_trace.csv> [delimiter-kernel-substring=adam]

Steps are cut at the launches of the delimiter kernel (the fused Adam ends a step).  For the last full step it prints the span, the
time during which 0 / 1 / 2+ kernels are running, per-queue busy time, and the kernels on the longest chain of back-to-back
(gap < 3 us) executions - the critical path as the GPU saw it."""
import csv
import re
import sys
from collections import defaultdict


def short(n):
    n = re.sub(r"\(anonymous namespace\)::", "", n)
    n = re.sub(r"_ZN12_GLOBAL__N_1\d+", "", n)
    n = re.sub(r"^void ", "", n)
    return n.split("(")[0][:60]


def main():
    path = sys.argv[1]
    delim = sys.argv[2] if len(sys.argv) > 2 else "adam"
    rows = []
    for r in csv.DictReader(open(path)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Queue_Id", "0"), r.get("Stream_Id", "0")))
    rows.sort()
    cuts = [i for i, r in enumerate(rows) if delim in r[2]]
    if len(cuts) < 3:
        print("not enough steps"); return
    lo, hi = cuts[-3] + 1, cuts[-2] + 1          # the second-to-last full step (the last may be cut by teardown)
    step = rows[lo:hi]
    t0 = min(r[0] for r in step); t1 = max(r[1] for r in step)
    print("step: %d kernels, span %.3f ms, sum of durations %.3f ms" % (len(step), (t1 - t0) / 1e6, sum(r[1] - r[0] for r in step) / 1e6))
    ev = []
    for s, e, *_ in step:
        ev.append((s, 1)); ev.append((e, -1))
    ev.sort()
    depth, last, hist = 0, t0, defaultdict(int)
    for t, d in ev:
        hist[min(depth, 3)] += t - last
        last = t; depth += d
    print("time with 0 / 1 / 2 / 3+ kernels running: " + " / ".join("%.3f" % (hist[k] / 1e6) for k in range(4)) + " ms")
    perq = defaultdict(lambda: [0, 0])
    for s, e, n, q, st in step:
        perq[(q, st)][0] += e - s; perq[(q, st)][1] += 1
    for k, v in sorted(perq.items(), key=lambda kv: -kv[1][0]):
        print("  queue %s stream %s: %4d kernels, busy %.3f ms" % (k[0], k[1], v[1], v[0] / 1e6))
    # time attributed to kernels while they run ALONE (depth 1) or as the longest-running of a group: who owns the wall clock
    own = defaultdict(float)
    active = []
    pts = sorted(set([r[0] for r in step] + [r[1] for r in step]))
    idx = 0
    import heapq
    step_sorted = sorted(step)
    live = []
    for a, b in zip(pts[:-1], pts[1:]):
        while idx < len(step_sorted) and step_sorted[idx][0] <= a:
            live.append(step_sorted[idx]); idx += 1
        live = [r for r in live if r[1] > a]
        if not live:
            own["<idle>"] += b - a
        else:
            for r in live:
                own[short(r[2])] += (b - a) / len(live)
    # idle intervals (nothing running): which kernel ended before and which starts after, summed by that pair of names
    gaps = defaultdict(lambda: [0, 0])
    biggest = []
    end_sorted = sorted(step, key=lambda r: r[1])
    cur_end, cur_name = None, None
    for r in step_sorted:
        if cur_end is not None and r[0] > cur_end:
            g = r[0] - cur_end
            gaps[(cur_name, short(r[2]))][0] += g; gaps[(cur_name, short(r[2]))][1] += 1
            biggest.append((g, (cur_end - t0) / 1e6, cur_name, short(r[2])))
        if cur_end is None or r[1] > cur_end:
            cur_end, cur_name = r[1], short(r[2])
    print("idle by (kernel that ended last -> kernel that starts): total us, count")
    for k, v in sorted(gaps.items(), key=lambda kv: -kv[1][0])[:25]:
        print("  %8.1f us %3d  %s -> %s" % (v[0] / 1e3, v[1], k[0][:44], k[1][:44]))
    print("largest idle intervals: us, at ms, between")
    for g, at, a, b in sorted(biggest, reverse=True)[:15]:
        print("  %8.1f us at %7.3f  %s -> %s" % (g / 1e3, at, a[:44], b[:44]))
    print("wall-clock share (time split equally among the kernels running at each instant):")
    for k, v in sorted(own.items(), key=lambda kv: -kv[1])[:28]:
        print("  %-62s %.3f ms" % (k, v / 1e6))


if __name__ == "__main__":
    main()
