"""Attention forward against torch fp32 at a few shapes: max abs error of ctx per (shape, ragged), per batch element."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
LOG2E = 1.4426950408889634
DEV = "cuda:0"
for (B, h, Lq, Lk, ragged) in [(2, 2, 300, 300, False), (3, 4, 200, 200, True), (2, 2, 256, 256, False), (2, 2, 200, 128, False), (2, 2, 200, 64, False), (2, 2, 200, 100, False), (1, 2, 1000, 1000, True), (2, 2, 200, 192, False)]:
    g = torch.Generator().manual_seed(B * 1000 + Lq)
    q = torch.randn(B, h, Lq, 64, generator=g) * 0.5
    k = torch.randn(B, h, Lk, 64, generator=g)
    v = torch.randn(B, h, Lk, 64, generator=g)
    k_len = None
    if ragged:
        k_len = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        k_len[0] = Lk
    qd, kd, vd = ((q * LOG2E).to(DEV).bfloat16(), k.to(DEV).bfloat16(), v.to(DEV).bfloat16())
    ctx, lse = ops.attention_fwd(qd, kd, vd, None if k_len is None else k_len.to(DEV).int(), False, need_lse=True)
    qr, kr, vr = (qd.float().cpu() / LOG2E, kd.float().cpu(), vd.float().cpu())
    s = qr @ kr.transpose(-1, -2)
    if k_len is not None:
        s = s.masked_fill((torch.arange(Lk)[None, :] >= k_len[:, None])[:, None, None, :], float("-inf"))
    p = torch.softmax(s, -1)
    ref = (p @ vr).permute(0, 2, 1, 3).reshape(B, Lq, h * 64)
    err = (ctx.float().cpu() - ref).abs()
    print((B, h, Lq, Lk, ragged), "k_len", None if k_len is None else k_len.tolist(), "max err per b:", [round(float(err[b].max()), 4) for b in range(B)],
          "lse err", round(float((lse.cpu() - torch.logsumexp(s, -1) * LOG2E).abs().max()), 4))
    if float(err.max()) > 0.05:
        bad = (err > 0.05).nonzero()
        print("   first bad (b, q, col):", bad[:3].tolist(), " bad q range", int(bad[:, 1].min()), int(bad[:, 1].max()), "count", len(bad))
