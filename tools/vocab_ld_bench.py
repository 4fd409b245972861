#!/usr/bin/env python3
"""ctc_fc projection [32000 x 256] x [4234 x 256]^T -> f32 logits: does the row stride of the logits (ldc) matter?  Rows of
4240 floats start 64 bytes off a 128-byte line every other row; 4256 = 133 * 32 floats keeps every row line-aligned."""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asr_amd import ops

DEV = "cuda:0"
B, L, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
x = torch.randn(B * L, 256, generator=g).to(DEV).bfloat16()
w = (torch.randn(V, 256, generator=g) * 0.1).to(DEV).bfloat16()
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), L, dtype=torch.int32).to(DEV)
for ld in (4240, 4256, 4288, 4352):
    buf = torch.empty((B * L, ld), device=DEV, dtype=torch.float32)
    res = {}
    for what in ("gemm", "ctc_fwd", "both"):
        ts = []
        for i in range(25):
            if what != "gemm":
                ops.gemm_nt_raw(x, B * L, 256, 256, w, None, out=buf, ldc=ld)
            a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            if what == "both":
                torch.cuda.synchronize()
            a.record()
            if what in ("gemm", "both"):
                ops.gemm_nt_raw(x, B * L, 256, 256, w, None, out=buf, ldc=ld)
            if what in ("ctc_fwd", "both"):
                ops.ctc_loss_fwd(buf[:, :V].view(B, L, V), il, tg)
            b.record()
            torch.cuda.synchronize()
            if i >= 5:
                ts.append(a.elapsed_time(b))
        ts.sort()
        res[what + "_ms"] = round(ts[len(ts) // 2], 4)
    print(json.dumps(dict(ldc=ld, **res)))
