#!/usr/bin/env python3
"""GPU time of the three segments of the eager S1 training step: encoder forward | decoder forward + losses + decoder backward |
encoder backward + Adam (events on the main stream at the two boundaries the trainer knows: the CTC fork and the join)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import asr_amd
import bench

dev = torch.device("cuda:0")
bench.CFG["n_conv_layers"] = int(os.environ.get("CONV", "0"))
model = bench.build_model(asr_amd, dev, 0.1, train=True)
asr_amd.manual_seed(1234)
x, lens, tg = bench.make_batch(dev, seed=0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
from asr_amd import modules as _m
for _k, _v in os.environ.items():      # MOD__CROSS_DKV_SIDE=0 -> modules._CROSS_DKV_SIDE = False (A/B of module constants)
    if _k.startswith("MOD_"):
        setattr(_m, _k[4:], bool(int(_v)))
tr.side_inline = os.environ.get("SIDE_INLINE") == "1"
for _k, _v in os.environ.items():      # TR_SIDE_BUDGET=128 -> tr.side_budget = 128
    if _k.startswith("TR_"):
        setattr(tr, _k[3:].lower(), int(_v))      # the CTC branch queued on the launch stream itself (no overlap with the decoder)
step = tr.step
for _ in range(6):
    step(x, lens, tg, max_target_len=bench.CFG["U"])
torch.cuda.synchronize()
rows = []
for _ in range(10):
    tr._seg_events = []
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    step(x, lens, tg, max_target_len=bench.CFG["U"])
    b.record()
    torch.cuda.synchronize()
    ev = dict(tr._seg_events)
    rows.append((a.elapsed_time(ev["enc_fwd_end"]), ev["enc_fwd_end"].elapsed_time(ev["enc_bwd_begin"]), ev["enc_bwd_begin"].elapsed_time(b),
                 a.elapsed_time(b)))
tr._seg_events = None
rows.sort(key=lambda r: r[3])
m = rows[len(rows) // 2]
print("encoder fwd %.2f ms | decoder fwd + loss + decoder bwd %.2f ms | encoder bwd + Adam %.2f ms | step %.2f ms" % m)
