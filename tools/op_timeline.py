#!/usr/bin/env python3
"""Time line of ONE eager S1 training step from HIP events around every C-ABI call (ops.profile_start(sequence=True)) - no profiler
attached, every stream on one time axis.  Prints the decoder segment (CTC fork .. join): each op's start, duration and stream, the
side streams' ops beside it, and the main-stream ops that ran > 2.5x their median.  SIDE_INLINE=1: the CTC branch on the launch stream.

    python tools/op_timeline.py [--all]
"""
import os
import sys
from collections import defaultdict

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import asr_amd
import bench
from asr_amd import ops

dev = torch.device("cuda:0")
model = bench.build_model(asr_amd, dev, 0.1, train=True)
asr_amd.manual_seed(1234)
x, lens, tg = bench.make_batch(dev, seed=0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
tr.side_inline = os.environ.get("SIDE_INLINE") == "1"
for k, v in os.environ.items():
    if k.startswith("TR_"):
        setattr(tr, k[3:].lower(), int(v))
for _ in range(6):
    tr.step(x, lens, tg, max_target_len=bench.CFG["U"])
torch.cuda.synchronize()
best = None
for _ in range(5):
    ops.profile_start(sequence=True)
    tr._seg_events = []
    t0 = ops.profile_mark()
    tr.step(x, lens, tg, max_target_len=bench.CFG["U"])
    t1 = ops.profile_mark()
    torch.cuda.synchronize()
    seq = ops.profile_sequence(t0)
    total = ops._elapsed_ms(t0, t1)
    ev = dict(tr._seg_events)
    ops.profile_stop()
    if best is None or total < best[0]:
        best = (total, seq, ev)
tr._seg_events = None
total, seq, ev = best
main = max(set(s for _, _, _, s in seq), key=lambda s: sum(1 for r in seq if r[3] == s))
names = {main: "main"}
for _, _, _, s in seq:
    names.setdefault(s, "side%d" % len(names))
print("step %.3f ms (bracketed: every op carries two timing events); %d ops, streams: %s" % (total, len(seq), ", ".join(sorted(names.values()))))
side = [r for r in seq if r[3] != main]
lo = min(r[1] for r in seq if "vocab_proj_ctc" in r[0] or "ctc_loss_fwd" in r[0])
hi = max((r[1] + r[2] for r in seq if r[3] == main and r[1] < lo + 3.2), default=lo)
seg = [r for r in seq if (lo - 0.05 <= r[1] <= lo + 3.0)] if "--all" not in sys.argv else seq
med = defaultdict(list)
for n, s, d, st in seq:
    if st == main:
        med[n].append(d)
med = {n: sorted(v)[len(v) // 2] for n, v in med.items()}
print("window: +%.3f ms .. +%.3f ms after the step's start (the CTC branch's first op .. 3 ms on)" % (lo, lo + 3.0))
for n, s, d, st in seg:
    flag = ""
    if st == main and d > 2.5 * med[n] and d > 0.03:
        beside = [m[:28] for m, s2, d2, st2 in side if s2 < s + d and s2 + d2 > s]
        flag = "   <-- %.1fx its median %.1f us; beside: %s" % (d / med[n], med[n] * 1e3, ", ".join(beside))
    print("  +%8.1f us  %7.1f us  %-6s %s%s" % (s * 1e3, d * 1e3, names[st], n, flag))
slow = sum(d - med[n] for n, s, d, st in seg if st == main and d > 2.5 * med[n] and d > 0.03)
print("main-stream time above the ops' medians in the window: %.3f ms" % slow)
