#!/usr/bin/env python3
"""The CTC branch of the S1 training step alone on the chip: plain ctc_fc GEMM + the streaming CTC forward (what a caller outside the
trainer gets) against the trainer's form - projection writing fp16 logits + lse + the CTC table rows (asr_vocab_proj_ctc), recursion
on the table - and both with the gradient pass behind them."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
dev = torch.device("cuda:0")
B, L, V, U = 32, 1000, 4234, 51
M = B * L
g = torch.Generator().manual_seed(0)
x = torch.randn(M, 256, generator=g).bfloat16().to(dev)
w = (torch.randn(V, 256, generator=g) * 0.1).bfloat16().to(dev)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(dev)
il = torch.full((B,), L, dtype=torch.int32, device=dev)
Vp = (V + 7) // 8 * 8
buf = torch.empty((M, Vp), device=dev, dtype=torch.float32)
one = torch.ones(1, device=dev)
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
lg = buf[:, :V].view(B, L, V)
def plain(bwd):
    ops.gemm_nt_raw(x, M, 256, 256, w, None, out=buf, ldc=Vp)
    _, _, st = ops.ctc_loss_fwd(lg, il, tg)
    if bwd: ops.ctc_loss_bwd(st, one, bf16=True)
def fused(bwd):
    _, _, _, st = ops.vocab_proj_ctc(x, w, tg, il, B, L)
    if bwd: ops.ctc_loss_bwd(st, one, bf16=True)
print("plain GEMM + streaming CTC forward: %.1f us forward, %.1f us with the gradient pass" % (timeit(lambda: plain(False)), timeit(lambda: plain(True))))
print("projection + lse + table rows (fp16 logits) + recursion: %.1f us forward, %.1f us with the gradient pass" % (timeit(lambda: fused(False)), timeit(lambda: fused(True))))
