#!/usr/bin/env python3
"""Where does the wall time of a training step go?  Reads a rocprofv3 --kernel-trace CSV of `bench.py`, cuts out whole steps
(adam_kernel marks a step's end), and reports per step: wall, time with NO kernel running on any queue (the GPU waiting for the
host or for a dependency), per-queue busy time, and the largest idle gaps with the kernels either side.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/tl -o tl -- python3 bench.py --steps 6 --warmup 2
    python tools/step_timeline.py gpurun_out/tl
"""
import csv
import glob
import json
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:48]


def main(d):
    f = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)[0]
    rows = []
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]))
    rows.sort()
    ends = [e for s, e, q, n in rows if n.startswith("adam_kernel") or "adam_kernel" in n[:40]]
    print(json.dumps(dict(kernels=len(rows), steps_seen=len(ends))))
    # steps k .. k+1 between consecutive adam ends; take the timed region of the FIRST pass (side streams on): bench runs 3 init +
    # warmup + steps, then a second eager pass; report steps 5..5+n
    out = []
    for k in range(4, min(len(ends) - 1, 10)):
        t0, t1 = ends[k], ends[k + 1]
        ks = [(s, e, q, n) for s, e, q, n in rows if s >= t0 and e <= t1 + 1]
        if not ks:
            continue
        ev = sorted([(s, 1) for s, e, q, n in ks] + [(e, -1) for s, e, q, n in ks])
        busy, depth, last = 0, 0, t0
        gaps = []
        cur_gap_start = t0
        for t, dlt in ev:
            if depth == 0 and dlt == 1:
                gaps.append((t - cur_gap_start, cur_gap_start, t))
            if depth > 0:
                busy += t - last
            depth += dlt
            last = t
            if depth == 0:
                cur_gap_start = t
        perq = defaultdict(int)
        for s, e, q, n in ks:
            perq[q] += e - s
        # idle by phase: small gaps (< 10 us) summed separately
        small = sum(g for g, a, b in gaps if g < 10000)
        big = sorted(gaps, reverse=True)[:6]
        def around(a, b):
            before = [n for s, e, q, n in ks if e == a]
            after = [n for s, e, q, n in ks if s == b]
            return (short(before[0]) if before else "-", short(after[0]) if after else "-")
        out.append(dict(step=k, wall_ms=(t1 - t0) / 1e6, busy_any_ms=busy / 1e6, idle_ms=(t1 - t0 - busy) / 1e6,
                        n_kernels=len(ks), n_gaps=len(gaps), idle_in_gaps_under_10us_ms=small / 1e6,
                        per_queue_busy_ms={q: round(v / 1e6, 3) for q, v in sorted(perq.items(), key=lambda x: -x[1])},
                        biggest_gaps_us=[(round(g / 1e3, 1),) + around(a, b) for g, a, b in big]))
    for o in out:
        print(json.dumps(o))
    # gap histogram of one step, attributed to the kernel that FOLLOWS the gap
    if out:
        k = out[len(out) // 2]["step"]
        t0, t1 = ends[k], ends[k + 1]
        ks = [(s, e, q, n) for s, e, q, n in rows if s >= t0 and e <= t1 + 1]
        # idle time on the busiest queue between consecutive kernels of that queue
        perq = defaultdict(list)
        for s, e, q, n in ks:
            perq[q].append((s, e, n))
        mainq = max(perq, key=lambda q: len(perq[q]))
        seq = sorted(perq[mainq])
        hist = defaultdict(lambda: [0, 0.0])
        for (s0, e0, n0), (s1, e1, n1) in zip(seq, seq[1:]):
            g = max(0, s1 - e0)
            h = hist[short(n1)]
            h[0] += 1
            h[1] += g / 1e3
        for q in perq:
            tot = defaultdict(lambda: [0, 0.0])
            for s_, e_, n_ in perq[q]:
                tot[short(n_)][0] += 1
                tot[short(n_)][1] += (e_ - s_) / 1e3
            print(json.dumps(dict(queue=q, kernel_us_per_step={k_: [v[0], round(v[1], 1)] for k_, v in sorted(tot.items(), key=lambda kv: -kv[1][1])[:40]})))
        top = sorted(hist.items(), key=lambda kv: -kv[1][1])[:14]
        print(json.dumps(dict(main_queue=mainq, launches=len(seq), gap_before_kernel_us_total={k: [v[0], round(v[1], 1)] for k, v in top},
                              total_gap_ms=round(sum(v[1] for v in hist.values()) / 1e3, 3),
                              busy_ms=round(sum(e - s for s, e, n in seq) / 1e6, 3))))


if __name__ == "__main__":
    main(sys.argv[1])
