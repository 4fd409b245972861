#!/usr/bin/env python3
"""ffn_fwd at the S1 shape, eval and train, 30 launches back to back (timing only: used by tools/abl_ffn.sh on ablated builds)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asr_amd import ops

dev = torch.device("cuda:0")
B, L, dff = 32, 1000, 2048
M = B * L
g = torch.Generator().manual_seed(0)
x32 = torch.randn(M, 256, generator=g).to(dev)
x16 = x32.bfloat16()
w1 = (torch.randn(dff, 256, generator=g) * 0.06).to(dev).bfloat16()
w2 = (torch.randn(256, dff, generator=g) * 0.03).to(dev).bfloat16()
b1 = torch.zeros(dff, device=dev)
b2 = torch.zeros(256, device=dev)
gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)


def t(train):
    fn = lambda: ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, train=train)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / 30 * 1e3


print("ffn_fwd eval %.1f us  train %.1f us" % (t(False), t(True)))
