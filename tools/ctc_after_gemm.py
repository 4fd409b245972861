#!/usr/bin/env python3
"""Does the CTC forward run slower right behind the GEMM that wrote its logits (as in the training step) than on logits that
have been resident for a while?  HIP-event time of asr_ctc_loss_fwd alone in both situations, contiguous and padded rows."""
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
Vp = (V + 7) // 8 * 8
buf = torch.empty((B * L, Vp), device=DEV, dtype=torch.float32)


def run(with_gemm, arena, iters=30):
    ts = []
    for i in range(iters + 3):
        if arena:
            ops.arena_reset(DEV)
        if with_gemm:
            ops.gemm_nt_raw(x, B * L, 256, 256, w, None, out=buf, ldc=Vp)
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record()
        ops.ctc_loss_fwd(buf[:, :V].view(B, L, V), il, tg)
        b.record()
        torch.cuda.synchronize()
        if arena:
            ops.arena_release()
        if i >= 3:
            ts.append(a.elapsed_time(b))
    ts.sort()
    return round(ts[len(ts) // 2], 4)


ops.gemm_nt_raw(x, B * L, 256, 256, w, None, out=buf, ldc=Vp)
for with_gemm in (False, True):
    for arena in (False, True):
        print(json.dumps(dict(gemm_before=with_gemm, arena_counters=arena, ctc_fwd_ms_median=run(with_gemm, arena))))
