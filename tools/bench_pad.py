"""Does a 4096-byte row stride (K = 2048 bf16) hurt the GEMMs that read the FFN hidden activation?  A with lda = 2048 vs 2112."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, asr_amd
from asr_amd import ops
DEV = "cuda:0"
M, D, F = 32000, 256, 2048
def t(fn, n=30):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
w2 = (torch.randn(D, F, device=DEV) * 0.02).bfloat16(); b2 = torch.zeros(D, device=DEV)
w1 = (torch.randn(F, D, device=DEV) * 0.05).bfloat16()
dy = torch.randn(M, D, device=DEV).bfloat16()
gw2 = torch.zeros(D, F, device=DEV)
for pad in (0, 64, 128):
    buf = torch.randn(M, F + pad, device=DEV).bfloat16()
    hid = buf[:, :F]
    us_nt = t(lambda: ops.gemm_nt_raw(buf, M, F, F + pad, w2, b2))                      # FFN2: A = hid
    us_tn = t(lambda: ops.gemm_tn(dy, hid, out=gw2, accumulate=True))                    # dW2: B = hid
    us_nn = t(lambda: ops.gemm_nn(hid, w1, lda=F + pad, K=F))                            # dX: A = d_hid (same shape)
    print("pad %3d: FFN2 nt %.1f us   dW2 tn %.1f us   dX nn %.1f us" % (pad, us_nt, us_tn, us_nn))
