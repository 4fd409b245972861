"""Tiny driver for counter collection: the S1 encoder FFN GEMMs (forward, dgrad, wgrad), 6 launches each."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
M, D, F = 32000, 256, 2048
x = torch.randn(M, D, device=DEV).bfloat16(); w1 = (torch.randn(F, D, device=DEV) * 0.05).bfloat16(); b1 = torch.zeros(F, device=DEV)
w2 = (torch.randn(D, F, device=DEV) * 0.02).bfloat16(); b2 = torch.zeros(D, device=DEV)
bits = torch.empty(M, F // 8, device=DEV, dtype=torch.uint8)
dy = torch.randn(M, D, device=DEV).bfloat16()
gw1 = torch.zeros(F, D, device=DEV); gw2 = torch.zeros(D, F, device=DEV); gb1 = torch.zeros(F, device=DEV)
for _ in range(6):
    hdn = ops.gemm_nt_ex(x, w1, b1, out_dtype=torch.bfloat16, relu=True, relu_bits_out=bits)      # FFN1
    y = ops.gemm_nt(hdn, w2, b2)                                                                   # FFN2
    dh = ops.gemm_nn(dy, w2, out_dtype=torch.bfloat16, relu_bits=bits)                             # hidden gradient
    dx = ops.gemm_nn(dh, w1)                                                                       # input gradient
    ops.gemm_tn(dy, hdn, out=gw2, accumulate=True)                                                 # dW2
    ops.gemm_tn(dh, x, out=gw1, accumulate=True, colsum=gb1)                                       # dW1
torch.cuda.synchronize()

if "--time" in sys.argv:
    def t(fn, n=30):
        for _ in range(5): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1000
    fl = 2.0 * M * D * F
    for name, fn in [("FFN1 nt relu bits bf16", lambda: ops.gemm_nt_ex(x, w1, b1, out_dtype=torch.bfloat16, relu=True, relu_bits_out=bits)),
                     ("FFN1 nt relu bf16", lambda: ops.gemm_nt_ex(x, w1, b1, out_dtype=torch.bfloat16, relu=True)),
                     ("FFN2 nt f32", lambda: ops.gemm_nt(hdn, w2, b2)),
                     ("dH nn bits bf16", lambda: ops.gemm_nn(dy, w2, out_dtype=torch.bfloat16, relu_bits=bits)),
                     ("dH nn bf16 (no mask)", lambda: ops.gemm_nn(dy, w2, out_dtype=torch.bfloat16)),
                     ("dX nn f32", lambda: ops.gemm_nn(dh, w1)),
                     ("dW2 tn", lambda: ops.gemm_tn(dy, hdn, out=gw2, accumulate=True)),
                     ("dW1 tn colsum", lambda: ops.gemm_tn(dh, x, out=gw1, accumulate=True, colsum=gb1))]:
        us = t(fn)
        print("%-26s %7.1f us  %6.0f TFLOP/s" % (name, us, fl / us / 1e6))
