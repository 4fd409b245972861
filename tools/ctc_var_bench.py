#!/usr/bin/env python3
"""CTC forward at the north-star shape under the A/B knobs of asr_ctc_loss_fwd (ASR_AMD_CTC_RPB / _RING / _DBG): HIP-event time
of the whole op call (workspace allocation, counter memset, fused kernel, mean)."""
import json
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))
import torch  # noqa: E402

from asr_amd import ops  # noqa: E402
from bench_ops import timeit  # noqa: E402

DEV = "cuda:0"
B, L, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
logits = torch.randn(B, L, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), L, dtype=torch.int32).to(DEV)
tag = {k: os.environ.get(k) for k in ("ASR_AMD_CTC_RPB", "ASR_AMD_CTC_RING", "ASR_AMD_CTC_DBG") if os.environ.get(k)}
for nck in [int(a) for a in sys.argv[1:]] or (1, 16, 32):
    t = timeit(lambda: ops.ctc_loss_fwd(logits, il, tg, n_chunks=nck), iters=40)
    print(json.dumps(dict(knobs=tag, n_chunks=nck, ms=round(t, 4))))
