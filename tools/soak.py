#!/usr/bin/env python3
"""Soak: N training steps of the benchmark model (S1 eager / S2 replayed), watching allocated / reserved memory and the losses.
A leak in the side-stream bookkeeping (kept operands, events, graph pools) or a drifting loss shows here, not in a 20-step bench."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import asr_amd
import bench

dev = torch.device("cuda:0")
bench.CFG["n_conv_layers"] = int(os.environ.get("CONV", "0"))
N = int(os.environ.get("STEPS", "400"))
model = bench.build_model(asr_amd, dev, 0.1, train=True)
asr_amd.manual_seed(7)
x, lens, tg = bench.make_batch(dev, seed=0, ragged=os.environ.get("RAGGED", "1") == "1")
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
rows = []
for i in range(N):
    ctc, ce = tr.step_auto(x, lens, tg, max_target_len=bench.CFG["U"])
    if i % (N // 8) == 0 or i == N - 1:
        torch.cuda.synchronize()
        rows.append((i, tr.launch_mode, float(ctc), float(ce), torch.cuda.memory_allocated() / 2**20, torch.cuda.memory_reserved() / 2**20))
for r in rows:
    print("step %4d  %-6s ctc %9.3f  ce %7.4f  allocated %8.1f MiB  reserved %8.1f MiB" % r)
a0, a1 = rows[2][4], rows[-1][4]
assert a1 <= a0 * 1.02 + 64, "allocated memory grows: %.1f -> %.1f MiB" % (a0, a1)
assert all(map(lambda r: r[2] == r[2] and r[3] == r[3], rows)), "NaN loss"
assert rows[-1][3] < rows[1][3], "CE loss did not go down"
print("soak ok")
