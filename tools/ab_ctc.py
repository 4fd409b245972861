#!/usr/bin/env python3
"""A/B of two builds of the library on the same box, same process order: ASR_LIB=<path> picks the .so (default: the product's)."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import asr_amd._lib as L

if os.environ.get("ASR_LIB"):
    L.LIB_PATH = os.path.abspath(os.environ["ASR_LIB"])
import torch

from asr_amd import ops

DEV = "cuda:0"
B, Lq, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
logits = torch.randn(B, Lq, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), Lq, dtype=torch.int32).to(DEV)
res = []
for rep in range(5):
    for _ in range(5):
        ops.ctc_loss_fwd(logits, il, tg)
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(50):
        ops.ctc_loss_fwd(logits, il, tg)
    b.record()
    torch.cuda.synchronize()
    res.append(round(a.elapsed_time(b) / 50, 4))
print(json.dumps(dict(lib=os.path.basename(L.LIB_PATH), ctc_fwd_ms=sorted(res))))
