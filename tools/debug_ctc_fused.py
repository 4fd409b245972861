#!/usr/bin/env python3
"""Where do the fused and the two-launch CTC forward differ?  (workspace rows, nll) - development aid.
usage: debug_ctc_fused.py <n_chunks>"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch

from asr_amd import ops

DEV = "cuda:0"
nck = int(sys.argv[1]) if len(sys.argv) > 1 else 2
B, L, U, V = 5, 330, 17, 91
g = torch.Generator().manual_seed(nck)
logits = torch.randn(B, L, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g)
tg[2, 9:] = 0
il = torch.tensor([330, 77, 201, 64, 1]).to(DEV)
tg[4, 1:] = 0
tg = tg.to(DEV)
l1, n1, s1 = ops.ctc_loss_fwd(logits, il, tg, n_chunks=1)
a1, lp1 = s1.alpha.clone(), s1.lp_ext.clone()
for rep in range(1):
    l2, n2, s2 = ops.ctc_loss_fwd(logits, il, tg, n_chunks=nck)
    a2, lp2 = s2.alpha.clone(), s2.lp_ext.clone()
    torch.cuda.synchronize()
    print("rep", rep, "nll", n1.cpu().numpy(), n2.cpu().numpy())
    for b in range(B):
        Tb = int(il[b])
        d = (lp1[b, :Tb] != lp2[b, :Tb]) & ~(torch.isnan(lp1[b, :Tb]) & torch.isnan(lp2[b, :Tb]))
        rows = list(range(Tb)) + [L]
        da = (a1[b, rows] != a2[b, rows]) & ~(torch.isnan(a1[b, rows]) & torch.isnan(a2[b, rows]))
        bad = da.any(-1).nonzero().flatten().tolist()
        good = [rows[r] for r in range(len(rows)) if r not in set(bad) and rows[r] > Tb // 2]
        print("  b", b, "Tb", Tb, "n bad", len(bad), "beta rows equal:", good[:12])
        if bad and rep == 0 and b == 3:
            U_ = int((tg[b] != 0).sum())
            for rr in (Tb - 4, Tb - 5, Tb - 6):
                cols = list(range(2 * U_ - 6, 2 * U_ + 2))
                print("    row", rr, "cols", cols)
                print("      ref", [round(v, 3) for v in a1[b, rr, cols].tolist()])
                print("      got", [round(v, 3) for v in a2[b, rr, cols].tolist()])
                print("      lp ", [round(v, 3) for v in lp1[b, rr, cols].tolist()], "lp[rr-1]", [round(v, 3) for v in lp1[b, rr - 1, cols].tolist()])
            import numpy as _np
            _np.savez("gpurun_out/ctc_dbg.npz", a1=a1[b].cpu().numpy(), a2=a2[b].cpu().numpy(), lp=lp1[b].cpu().numpy(), tg=tg[b].cpu().numpy(), Tb=Tb)
        if False:
            r = rows[bad[-1]]
            print("    last bad row", r, "ref", a1[b, r, :8].tolist(), "got", a2[b, r, :8].tolist())
            r = rows[bad[-2]] if len(bad) > 1 else r
            print("    prev bad row", r, "ref", a1[b, r, :8].tolist(), "got", a2[b, r, :8].tolist())
