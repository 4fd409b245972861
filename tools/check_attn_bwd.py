"""Attention backward (dq + dkv through the C-ABI) against float64 autograd, eval and with dropout: error statistics per output.
usage: python tools/check_attn_bwd.py [B h Lq Lk]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import asr_amd
from asr_amd import ops
from oracle import asr_oracle as O
LOG2E = 1.4426950408889634
DEV = "cuda:0"
B, h, Lq, Lk = (int(x) for x in sys.argv[1:5]) if len(sys.argv) > 4 else (2, 4, 200, 200)
THR = 6554
for drop in (False, True):
    g = torch.Generator().manual_seed(Lq * 3 + Lk)
    qdev = (torch.randn(B, h, Lq, 64, generator=g) * 0.4 * LOG2E).bfloat16()
    q = (qdev.double() / LOG2E).requires_grad_(True)
    k = torch.randn(B, h, Lk, 64, generator=g).bfloat16().double().requires_grad_(True)
    v = torch.randn(B, h, Lk, 64, generator=g).bfloat16().double().requires_grad_(True)
    k_len = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g); k_len[0] = Lk
    mask = (torch.arange(Lk)[None, :] >= k_len[:, None])[:, None, None, :]
    p = torch.softmax((q @ k.transpose(-1, -2)).masked_fill(mask, float("-inf")), -1)
    if drop:
        dm = torch.from_numpy(O.dropout_mask((h * B, Lq, Lk), THR, 31, 32)).view(h, B, Lq, Lk).permute(1, 0, 2, 3).double()
        p = p * dm
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, Lq, h * 64)
    dctx = torch.randn(B, Lq, h * 64, generator=g).bfloat16().double()
    ctx.backward(dctx)
    d = ops.Dropout(THR, 31, 32) if drop else None
    qd, kd, vd = qdev.to(DEV), k.detach().float().to(DEV).bfloat16(), v.detach().float().to(DEV).bfloat16()
    kl = k_len.to(DEV).int()
    ctx_d, lse = ops.attention_fwd(qd, kd, vd, kl, False, need_lse=True, drop=d)
    dq = torch.zeros(B * Lq, h * 64, device=DEV, dtype=torch.bfloat16)
    dkv = torch.zeros(B * Lk, 2 * h * 64, device=DEV, dtype=torch.bfloat16)
    ops.attention_bwd(qd, kd, vd, ctx_d, dctx.float().to(DEV).bfloat16(), lse, kl, False, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:], drop=d)
    torch.cuda.synchronize()
    to_tok = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0] * t.shape[2], h * 64)
    for name, got, ref in (("dq", dq, to_tok(q.grad) * 0.125), ("dk", dkv[:, :h * 64], to_tok(k.grad)), ("dv", dkv[:, h * 64:], to_tok(v.grad))):
        e = (got.double().cpu() - ref).abs()
        bad = (e > 5e-2 + 3e-2 * ref.abs())
        rows = bad.any(1).nonzero().flatten()
        print("drop=%d %s: max err %.4f  rel-L2 %.4f  violations %d of %d  rows %s  nan %d" % (
            drop, name, float(e.max()), float((got.double().cpu() - ref).norm() / ref.norm()), int(bad.sum()), bad.numel(), rows[:12].tolist(), int(torch.isnan(got).sum())))
