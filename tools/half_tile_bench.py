import os, sys
sys.path.insert(0, os.getcwd())
import torch
from asr_amd import ops
DEV="cuda:0"
M=8000
def t(fn, n=50):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    a,b=torch.cuda.Event(enable_timing=True),torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n): fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b)/n*1000
h=torch.randn(M,2048,device=DEV).bfloat16(); w2=(torch.randn(256,2048,device=DEV)*0.02).bfloat16(); b2=torch.zeros(256,device=DEV)
x=torch.randn(M,256,device=DEV).bfloat16(); wfc=(torch.randn(256,256,device=DEV)*0.05).bfloat16()
w1=(torch.randn(2048,256,device=DEV)*0.05).bfloat16(); dh=torch.randn(M,2048,device=DEV).bfloat16(); ds=torch.randn(M,256,device=DEV)
print("half=%s" % os.environ.get("ASR_AMD_HALF_M","1"),
      "FFN2 nt [8000x256x2048] %.1f us" % t(lambda: ops.gemm_nt(h,w2,b2)),
      "| fc nt [8000x256x256] %.1f us" % t(lambda: ops.gemm_nt(x,wfc,b2)),
      "| dX nn [8000x256x2048] %.1f us" % t(lambda: ops.gemm_nn(dh,w1,addend=ds)),
      "| dctx nn [8000x256x256] %.1f us" % t(lambda: ops.gemm_nn(x,wfc,out_dtype=torch.bfloat16)))
