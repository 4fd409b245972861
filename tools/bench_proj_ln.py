"""asr_proj_ln_fwd (attention out-projection + dropout + residual + LayerNorm in one launch) against the GEMM + LayerNorm pair at S1's encoder size."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, L = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1000
M = B * L
ctx = torch.randn(M, 256, device=DEV).bfloat16(); res = torch.randn(M, 256, device=DEV)
w = (torch.randn(256, 256, device=DEV) * 0.06).bfloat16(); bias = torch.randn(256, device=DEV) * 0.1
gam = torch.ones(256, device=DEV); bet = torch.zeros(256, device=DEV)
lens = torch.full((B,), L, device=DEV, dtype=torch.int32)


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for train in (False, True):
    d = ops.Dropout(6554, 7, 3) if train else None
    def pair():
        o = ops.gemm_nt(ctx, w, bias)
        return ops.add_layernorm(o, res, gam, bet, B, L, row_len=lens, want_bf16=True, save_stats=train, drop_x=d)
    def fused():
        return ops.proj_ln(ctx, w, bias, res, gam, bet, B, L, row_len=lens, save_stats=train, drop_x=d)
    print("train" if train else "eval", "pair %.1f us   fused %.1f us" % (t(pair), t(fused)))
