#!/usr/bin/env python3
"""ctc_fc projection + CTC forward at the S1 shape: plain GEMM + fused CTC forward (streams the logits) against the projection that takes
the row lse itself + the gather / recursion forward."""
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
def timeit(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3
t_gemm = timeit(lambda: ops.gemm_nt_raw(x, M, 256, 256, w, None, out=buf, ldc=Vp))
lg = buf[:, :V].view(B, L, V)
t_ctc = timeit(lambda: ops.ctc_loss_fwd(lg, il, tg))
t_both = timeit(lambda: (ops.gemm_nt_raw(x, M, 256, 256, w, None, out=buf, ldc=Vp), ops.ctc_loss_fwd(lg, il, tg)))
print("GEMM %.1f us | fused CTC forward %.1f us | back to back %.1f us" % (t_gemm, t_ctc, t_both))
t_v = timeit(lambda: ops.vocab_proj_lse(x, w))
logits, lse = ops.vocab_proj_lse(x, w)
l3 = logits.view(B, L, V)
t_c2 = timeit(lambda: ops.ctc_loss_fwd(l3, il, tg, lse=lse))
t_b2 = timeit(lambda: ops.ctc_loss_fwd(ops.vocab_proj_lse(x, w)[0].view(B, L, V), il, tg, lse=lse))
print("projection + lse %.1f us (%.0f TF, %.2f TB/s written) | CTC forward from lse %.1f us (%.2f TB/s of the unfused 542 MB) | back to back %.1f us" % (
    t_v, 2.0 * M * V * 256 / t_v / 1e6, M * Vp * 4 / t_v / 1e6, t_c2, 4.0 * M * V / t_c2 / 1e6, t_b2))
# round 5: bf16 logits + lse + the CTC table rows from the projection's own launch, recursion on the table; and both backward passes
t_c3 = timeit(lambda: ops.vocab_proj_ctc(x, w, tg, il, B, L))
one = torch.ones(1, device=dev)


def fwd_bwd_new():
    _, _, _, st = ops.vocab_proj_ctc(x, w, tg, il, B, L)
    ops.ctc_loss_bwd(st, one, bf16=True)


def fwd_bwd_old():
    lg2, lse2 = ops.vocab_proj_lse(x, w)
    _, _, st = ops.ctc_loss_fwd(lg2.view(B, L, V), il, tg, lse=lse2)
    ops.ctc_loss_bwd(st, one, bf16=True)


print("projection + lse + table (bf16 logits) + recursion %.1f us | with the gradient pass: %.1f us (round 4's form: %.1f us)" % (
    t_c3, timeit(fwd_bwd_new), timeit(fwd_bwd_old)))
