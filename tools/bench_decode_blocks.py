#!/usr/bin/env python3
"""Event-timed one-launch decode sub-layers (decode_blocks.hip) against row count, back-to-back launches on one stream."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asr_amd import ops

dev = torch.device("cuda:0")
g = torch.Generator().manual_seed(0)
r = lambda *s, sc=1.0: (torch.randn(*s, generator=g) * sc).to(dev)
w1, b1, w2, b2 = r(2048, 256, sc=1 / 16).bfloat16(), r(2048, sc=0.1), r(256, 2048, sc=1 / 45).bfloat16(), r(256, sc=0.1)
wqkv, bqkv, wo, bo = r(768, 256, sc=1 / 16).bfloat16(), r(768, sc=0.1), r(256, 256, sc=1 / 16).bfloat16(), r(256, sc=0.1)
gamma, beta = torch.ones(256, device=dev), torch.zeros(256, device=dev)
for M in (16, 32, 64, 96, 160, 256, 512):
    x = r(M, 256)
    xb = x.bfloat16()
    kc, vc = r(M, 4, 64, 64).bfloat16(), r(M, 4, 64, 64).bfloat16()
    state = torch.tensor([50, -1], dtype=torch.int32, device=dev)
    res = {"M": M}
    for name, fn in (("ffn", lambda: ops.decode_ffn(xb, x, w1, b1, w2, b2, gamma, beta, 1e-5)),
                     ("ffn_pre", lambda: ops.decode_ffn(xb, x, w1, b1, w2, b2, gamma, beta, 1e-5, pre=(wo, bo, gamma, beta, 1e-5))),
                     ("self_attn_t50", lambda: ops.decode_self_attn(xb, x, wqkv, bqkv, wo, bo, gamma, beta, kc, vc, state, 1e-5))):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res[name + "_us"] = round(e0.elapsed_time(e1) * 1e3 / 50, 2)
    print(json.dumps(res))
