"""Tiny driver for counter collection: a few launches of the fused feed-forward forward (train / eval) and backward at the S1 shape."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, L, dff = 32, 1000, 2048
M = B * L
torch.manual_seed(0)
x32 = torch.randn(M, 256, device=DEV); x16 = x32.bfloat16()
w1 = (torch.randn(dff, 256, device=DEV) * 0.06).bfloat16(); w2 = (torch.randn(256, dff, device=DEV) * 0.03).bfloat16()
b1 = torch.randn(dff, device=DEV) * 0.1; b2 = torch.randn(256, device=DEV) * 0.1
gamma = torch.ones(256, device=DEV); beta = torch.zeros(256, device=DEV)
lens = torch.full((B,), L, device=DEV, dtype=torch.int32)
drop = ops.Dropout(6554, 5, 9, None)
ds32 = torch.randn(M, 256, device=DEV) * 0.01; ds16 = ds32.bfloat16()
for _ in range(5):
    f = ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=True, drop_x=drop)
for _ in range(5):
    ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=False)
for _ in range(5):
    ops.ffn_bwd(ds16, ds32, w1, w2, f[1])
torch.cuda.synchronize()
