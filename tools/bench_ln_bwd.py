#!/usr/bin/env python3
"""add_layernorm_bwd alone on the chip at the encoder shapes of the S1 model (32000 x 256) and of the aishell recipe (10688 x 512)."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asr_amd import ops

dev = torch.device("cuda:0")
for B, L, D in ((32, 1000, 256), (32, 334, 512), (32, 51, 256), (32, 51, 512)):
    M = B * L
    g = torch.Generator().manual_seed(0)
    dy = torch.randn(M, D, generator=g).to(dev)
    s = torch.randn(M, D, generator=g).to(dev)
    mean, rstd = torch.randn(M, generator=g).to(dev), torch.rand(M, generator=g).to(dev) + 0.5
    gam = torch.randn(D, generator=g).to(dev)
    dg, db, dbias = (torch.zeros(D, device=dev) for _ in range(3))
    fn = lambda: ops.add_layernorm_bwd(dy, s, mean, rstd, gam, None, B, L, dg, db, want_bf16=True, dbias=dbias)
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30):
        fn()
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 30 * 1e3
    print("add_layernorm_bwd [%d x %d]: %.1f us = %.2f TB/s" % (M, D, us, 14.0 * M * D / us / 1e6))
