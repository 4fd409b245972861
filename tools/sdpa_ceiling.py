#!/usr/bin/env python3
"""Yardstick only (the product never calls it): torch's scaled_dot_product_attention (AOTriton / CK flash kernels on ROCm) on the
encoder self-attention shape, forward and forward + backward, against this repo's kernels on the same tensors."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import torch.nn.functional as F

from asr_amd import ops

dev = "cuda:0"
B, h, L = 32, 4, 1000
q = (torch.randn(B, h, L, 64, device=dev) * 0.7).bfloat16().requires_grad_(True)
k = torch.randn(B, h, L, 64, device=dev).bfloat16().requires_grad_(True)
v = torch.randn(B, h, L, 64, device=dev).bfloat16().requires_grad_(True)


def t(fn, n=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


base = 4.0 * B * h * 64 * L * L
res = {}
for name, ctx in (("flash", torch.nn.attention.SDPBackend.FLASH_ATTENTION), ("efficient", torch.nn.attention.SDPBackend.EFFICIENT_ATTENTION)):
    try:
        with torch.nn.attention.sdpa_kernel(ctx):
            with torch.no_grad():
                us = t(lambda: F.scaled_dot_product_attention(q, k, v))
            res[name + "_fwd_us"] = round(us, 1)
            res[name + "_fwd_TF"] = round(base / us / 1e6, 1)
            o = F.scaled_dot_product_attention(q, k, v)
            g = torch.randn_like(o)

            def fb():
                o = F.scaled_dot_product_attention(q, k, v)
                o.backward(g)
            us2 = t(fb)
            res[name + "_fwd_bwd_us"] = round(us2, 1)
            res[name + "_bwd_us"] = round(us2 - us, 1)
    except Exception as e:
        res[name + "_error"] = "%s: %s" % (type(e).__name__, str(e)[:120])
qd, kd, vd = q.detach(), k.detach(), v.detach()
us = t(lambda: ops.attention_fwd(qd, kd, vd, None, False, need_lse=True))
res["repo_fwd_us"] = round(us, 1)
res["repo_fwd_TF"] = round(base / us / 1e6, 1)
ctx, lse = ops.attention_fwd(qd, kd, vd, None, False, need_lse=True)
dctx = torch.randn_like(ctx)
dq = torch.empty(B * L, h * 64, device=dev, dtype=torch.bfloat16)
dkv = torch.empty(B * L, 2 * h * 64, device=dev, dtype=torch.bfloat16)
us = t(lambda: ops.attention_bwd(qd, kd, vd, ctx, dctx, lse, None, False, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:]))
res["repo_bwd_us"] = round(us, 1)
print(json.dumps(res))
