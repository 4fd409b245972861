"""Encoder attention forward (the v4 kernel's case: non-causal, Lq >= 128) against an fp64 softmax at ordinary and at EXTREME score
ranges (the re-centring branch of attention_fwd4.hip must fire: scores far outside [-64, 64] in base 2, rows whose first tiles lie
hundreds below their maximum, rows that start hundreds above zero).  Prints one line per case; exit code 1 on failure."""
import os
import sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
LOG2E = 1.4426950408889634
DEV = "cuda:0"
bad = 0
cases = [(2, 2, 300, 300, False, 1.0, "plain"), (3, 4, 200, 200, True, 1.0, "plain"), (2, 4, 1000, 1000, True, 1.0, "plain"), (1, 8, 256, 64, False, 1.0, "plain"),
         (2, 2, 130, 1, False, 1.0, "plain"), (2, 2, 257, 5, True, 1.0, "plain"), (2, 2, 384, 129, False, 1.0, "plain"),
         (2, 2, 300, 300, False, 12.0, "wide"), (2, 4, 1000, 1000, True, 30.0, "wide"), (2, 2, 300, 300, False, 1.0, "ramp-up"),
         (2, 2, 300, 300, True, 1.0, "ramp-down"), (2, 2, 512, 512, False, 1.0, "low"), (2, 2, 512, 512, False, 1.0, "high"), (1, 2, 256, 640, False, 1.0, "spike")]
for (B, h, Lq, Lk, ragged, scale, kind) in cases:
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk)
    q = torch.randn(B, h, Lq, 64, generator=g) * 0.5 * scale
    k = torch.randn(B, h, Lk, 64, generator=g)
    v = torch.randn(B, h, Lk, 64, generator=g)
    if kind in ("ramp-up", "ramp-down", "low", "high", "spike"):
        # dimension 0 carries an additive per-key offset: q[..., 0] = 1 (natural units), k[..., 0] = offset
        q[..., 0] = 1.0
        pos = torch.arange(Lk, dtype=torch.float32)
        off = {"ramp-up": pos * 1.5 - 300.0, "ramp-down": 250.0 - pos * 1.5, "low": torch.full((Lk,), -400.0), "high": torch.full((Lk,), 300.0),
               "spike": torch.where(pos == 333, torch.tensor(500.0), torch.tensor(-100.0))}[kind]
        k[..., 0] = off
    k_len = None
    if ragged:
        k_len = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        k_len[0] = Lk
        if B > 1:
            k_len[1] = min(Lk, 3)
    qd, kd, vd = ((q * LOG2E).to(DEV).bfloat16(), k.to(DEV).bfloat16(), v.to(DEV).bfloat16())
    ctx, lse = ops.attention_fwd(qd, kd, vd, None if k_len is None else k_len.to(DEV).int(), False, need_lse=True)
    torch.cuda.synchronize()
    qr, kr, vr = (qd.double().cpu(), kd.double().cpu(), vd.double().cpu())
    s = qr @ kr.transpose(-1, -2)          # base-2 logits
    if k_len is not None:
        s = s.masked_fill((torch.arange(Lk)[None, :] >= k_len[:, None])[:, None, None, :], float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    l = p.sum(-1, keepdim=True)
    ref = ((p / l) @ vr).permute(0, 2, 1, 3).reshape(B, Lq, h * 64)
    lse_ref = (m + torch.log2(l)).squeeze(-1)
    err = float((ctx.double().cpu() - ref).abs().max())
    lerr = float(((lse.double().cpu() - lse_ref).abs() / lse_ref.abs().clamp(min=1.0)).max())
    ok = err < 0.03 and lerr < 2e-3 and bool(torch.isfinite(ctx).all())
    bad += 0 if ok else 1
    print("%-9s B%d h%d %4dx%-4d ragged=%d scale=%-4g  |s|max %7.1f  ctx err %.4f  lse rel err %.2e  %s" %
          (kind, B, h, Lq, Lk, ragged, scale, float(s[torch.isfinite(s)].abs().max()), err, lerr, "ok" if ok else "FAIL"))
sys.exit(1 if bad else 0)
