#!/usr/bin/env python3
"""When does each workgroup of the fused CTC forward start and end?  ASR_AMD_CTC_DBG=128 makes every workgroup stamp the 100 MHz counter
into the spare workspace row; prints the end-time distribution of the pass workgroups and of the recursion workgroups (us after the first
start), per XCD (blockIdx mod 8)."""
import os
import sys

os.environ["ASR_AMD_CTC_DBG"] = str(128 | int(os.environ.get("ASR_AMD_CTC_DBG", "0")))
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from asr_amd import ops

DEV = "cuda:0"
B, L, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
logits = torch.randn(B, L, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), L, dtype=torch.int32).to(DEV)
for _ in range(5):
    loss, nll, st = ops.ctc_loss_fwd(logits, il, tg)
torch.cuda.synchronize()
for rep in range(3):
    loss, nll, st = ops.ctc_loss_fwd(logits, il, tg)
    torch.cuda.synchronize()
    rows = st.alpha[:, L + 1, :].contiguous().view(torch.int32).cpu().numpy().astype(np.uint32).reshape(-1, 2)[:2048]
    # word 0: HW_ID[15:0] (wave 3:0, simd 5:4, pipe 7:6, cu 11:8, sh 12, se 15:13) | XCC_ID << 16;  word 1: end time (100 MHz)
    hw = rows[:, 0]
    cu = ((hw >> 16) & 0xf) * 1000 + ((hw >> 13) & 7) * 100 + ((hw >> 12) & 1) * 50 + ((hw >> 8) & 0xf)
    end = (rows[:, 1] - rows[:, 1].min()).astype(np.int64) / 100.0
    rec, pas = slice(0, B), slice(B, 2048)
    q = lambda a: " ".join("%6.1f" % v for v in np.percentile(a, [0, 10, 50, 90, 99, 100]))
    chain_cus = set(cu[rec].tolist())
    on_chain_cu = np.array([c in chain_cus for c in cu[pas]])
    print("rep %d: %d distinct CUs; recursion workgroups on %d CUs; XCC of blockIdx 0..15: %s" % (rep, len(set(cu.tolist())), len(chain_cus), ((hw[:16] >> 16) & 0xf).tolist()))
    print("        pass end (us after the first end) p0/10/50/90/99/100: %s" % q(end[pas]))
    print("          ... on CUs that host a recursion workgroup (%d): %s" % (on_chain_cu.sum(), q(end[pas][on_chain_cu])))
    print("          ... on the other CUs (%d)                     : %s" % ((~on_chain_cu).sum(), q(end[pas][~on_chain_cu])))
    print("        chain end                                      : %s" % q(end[rec]))
    per_cu = {}
    for c, e in zip(cu[pas].tolist(), end[pas].tolist()):
        per_cu.setdefault(c, []).append(e)
    n_per = np.array([len(v) for v in per_cu.values()])
    print("        pass workgroups per CU: min %d max %d; CU-max end p0/50/100: %s" % (n_per.min(), n_per.max(), " ".join("%.1f" % v for v in np.percentile([max(v) for v in per_cu.values()], [0, 50, 100]))))
