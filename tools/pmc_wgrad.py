"""A few slab weight-gradient launches at the FFN shapes with 256 and 128 workgroups (run under rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
for (M, N, K) in ((32000, 2048, 256), (32000, 256, 2048)):
    a = torch.randn(M, N, device=DEV).bfloat16(); b = torch.randn(M, K, device=DEV).bfloat16()
    for wgs in (256, 128):
        out = torch.empty(N, K, device=DEV)
        for _ in range(5):
            ops.gemm_tn(a, b, out=out, max_wgs=wgs)
torch.cuda.synchronize()
