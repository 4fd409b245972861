"""Tiny driver for counter collection: 6 launches of the S1 encoder self-attention forward (and backward with --bwd)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, h, L = 32, 4, 1000
q = (torch.randn(B, h, L, 64, device=DEV) * 0.7).bfloat16(); k = torch.randn(B, h, L, 64, device=DEV).bfloat16(); v = torch.randn(B, h, L, 64, device=DEV).bfloat16()
drop = ops.Dropout(6554, 1, 2) if "--drop" in sys.argv else None
bits = ops.attention_dropmask(drop, B, h, L, L, DEV) if drop else None
for _ in range(6):
    ctx, lse = ops.attention_fwd(q, k, v, None, False, need_lse=True, drop=drop, drop_bits=bits)
if "--bwd" in sys.argv:
    dctx = torch.randn_like(ctx)
    dq = torch.empty(B * L, h * 64, device=DEV, dtype=torch.bfloat16); dkv = torch.empty(B * L, 2 * h * 64, device=DEV, dtype=torch.bfloat16)
    for _ in range(4):
        ops.attention_bwd(q, k, v, ctx, dctx, lse, None, False, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:], drop=drop, drop_bits=bits)
torch.cuda.synchronize()
