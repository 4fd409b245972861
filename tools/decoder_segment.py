#!/usr/bin/env python3
"""The decoder segment of a replayed S1 training step, kernel by kernel: from the end of the encoder's last forward kernel to the start of
the encoder's first backward kernel on the busiest queue, out of a rocprofv3 --kernel-trace CSV of bench.py.

    rocprofv3 --kernel-trace --output-format csv -d gpurun_out/dec -- python3 bench.py --steps 6 --no-cpu-baseline
    python tools/decoder_segment.py gpurun_out/dec
"""
import csv
import glob
import sys
from collections import defaultdict


def short(n):
    n = n.replace("void (anonymous namespace)::", "").replace("(anonymous namespace)::", "")
    return n.split("(")[0][:56]


def main(d):
    best = None
    for f in glob.glob(d + "/**/*kernel_trace.csv", recursive=True):
        rows = [(int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r.get("Queue_Id", "0"), r["Kernel_Name"]) for r in csv.DictReader(open(f))]
        if sum("ffn_bwd_kernel" in r[3] for r in rows) > (0 if best is None else sum("ffn_bwd_kernel" in r[3] for r in best)):
            best = rows      # (bench.py's child processes - the `also` legs - leave traces of their own)
    rows = sorted(best)
    ends = [e for s, e, q, n in rows if "adam_dev_kernel" in n]
    fwd, bwd = [], []
    for k in range(len(ends) // 2, len(ends) - 1):      # a replayed step in the middle of the first (timed) pass
        t0, t1 = ends[k], ends[k + 1]
        ks = [r for r in rows if r[0] >= t0 and r[1] <= t1 + 1]
        perq = defaultdict(int)
        for s, e, q, n in ks:
            if "ffn_bwd_kernel" in n:
                perq[q] += 1
        if not perq:
            continue
        mq = max(perq, key=perq.get)
        main_k = [r for r in ks if r[2] == mq]
        fwd = [i for i, r in enumerate(main_k) if "ffn_fwd2_kernel" in r[3]]      # (the encoder's feed-forward forward: ffn2.hip)
        bwd = [i for i, r in enumerate(main_k) if "ffn_bwd_kernel" in r[3]]
        if len(fwd) >= 12 and bwd:
            break
    a, b = fwd[-1], bwd[0]
    seg = main_k[a:b + 1]
    span = seg[-1][0] - seg[0][1]
    print("step wall %.3f ms; decoder segment (last encoder ffn_fwd end -> first ffn_bwd start) %.3f ms, %d kernels on the main queue" %
          ((t1 - t0) / 1e6, span / 1e6, len(seg) - 2))
    busy = sum(e - s for s, e, q, n in seg[1:-1])
    print("  main-queue busy %.3f ms, gaps %.3f ms" % (busy / 1e6, (span - busy) / 1e6))
    other = [r for r in ks if r[2] != mq and r[1] > seg[0][1] and r[0] < seg[-1][0]]
    print("  other queues inside the segment: %d kernels, %.3f ms busy" % (len(other), sum(min(e, seg[-1][0]) - max(s, seg[0][1]) for s, e, q, n in other) / 1e6))
    agg = defaultdict(lambda: [0, 0])
    for s, e, q, n in seg[1:-1]:
        agg[short(n)][0] += 1
        agg[short(n)][1] += e - s
    for n, (c, t) in sorted(agg.items(), key=lambda kv: -kv[1][1])[:28]:
        print("  %-58s x%-3d %7.1f us  (%.1f each)" % (n, c, t / 1e3, t / 1e3 / c))
    gaps = sorted(((seg[i + 1][0] - seg[i][1], short(seg[i][3]), short(seg[i + 1][3])) for i in range(len(seg) - 1)), reverse=True)[:10]
    print("  largest gaps:")
    for g, x, y in gaps:
        print("    %6.1f us  %s -> %s" % (g / 1e3, x, y))
    if "--list" in sys.argv:
        for i in range(1, len(seg) - 1):
            s, e, q, n = seg[i]
            print("    +%8.1f us  %6.1f us  gap %5.1f  %s" % ((s - seg[0][1]) / 1e3, (e - s) / 1e3, (s - seg[i - 1][1]) / 1e3, short(n)))
    if "--side" in sys.argv:          # the other queues' kernels on the same axis, and the main-queue kernels that ran > 3x their median beside them
        print("  other queues (start, duration, queue):")
        for s, e, q, n in sorted(other):
            print("    +%8.1f us  %6.1f us  q%-3s %s" % ((s - seg[0][1]) / 1e3, (e - s) / 1e3, q, short(n)))
        med = defaultdict(list)
        for s, e, q, n in seg[1:-1]:
            med[short(n)].append(e - s)
        print("  main-queue kernels at > 3x their median, and what ran beside them:")
        for s, e, q, n in seg[1:-1]:
            m = sorted(med[short(n)])[len(med[short(n)]) // 2]
            if e - s > 3 * m:
                beside = [short(n2)[:40] for s2, e2, q2, n2 in other if s2 < e and e2 > s]
                print("    +%8.1f us  %6.1f us (median %.1f)  %s   | %s" % ((s - seg[0][1]) / 1e3, (e - s) / 1e3, m / 1e3, short(n), ", ".join(beside)))


if __name__ == "__main__":
    main(sys.argv[1])
