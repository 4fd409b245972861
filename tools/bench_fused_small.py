import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, asr_amd
from asr_amd import ops
DEV = "cuda:0"
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3
for K in (256, 2048):
    for M in (16, 160, 1632, 4096):
        a = torch.randn(M, K, device=DEV).bfloat16(); w = (torch.randn(256, K, device=DEV) / K ** 0.5).bfloat16()
        bias = torch.randn(256, device=DEV); res = torch.randn(M, 256, device=DEV); g = torch.ones(256, device=DEV); be = torch.zeros(256, device=DEV)
        us = t(lambda: ops.gemm_add_layernorm_small(a, w, bias, res, g, be, 1, M, save_stats=True))
        def unf():
            o = ops.gemm_nt(a, w, bias); ops.add_layernorm(o, res, g, be, 1, M, want_bf16=True, save_stats=True)
        print("K=%4d M=%5d fused %6.1f us   unfused %6.1f us" % (K, M, us, t(unf)))
