"""CTC gradient error at the north-star shape (B 32, T 1000, U 50, V 4234), per utterance: this kernel (fp32, base-2 log domain) and aten's
fp32 `F.log_softmax + F.ctc_loss` against aten in float64 - the numbers behind the bounds of
tests/test_gpu_fullsize.py::test_ctc_full_size_against_aten_all_utterances."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch

import asr_amd

F = torch.nn.functional
B, T, U, V = 32, 1000, 50, 4234
DEV = "cuda:0"
for repeats in (False, True):
    g = torch.Generator().manual_seed(5 + int(repeats))
    logits = torch.randn(B, T, V, generator=g)
    tg = torch.randint(1, V - 1, (B, U), generator=g)
    if repeats:
        rep = torch.rand(B, U, generator=g) < 0.2
        rep[:, 0] = False
        for u in range(1, U):
            tg[:, u] = torch.where(rep[:, u], tg[:, u - 1], tg[:, u])
    tg[3, 30:] = 0
    tg[7, 1:] = 0
    il = torch.randint(2 * U + 2, T + 1, (B,), generator=g)
    il[0], il[1] = T, T
    tl = tg.ne(0).int().sum(1)

    def aten(dtype):
        lg = logits.detach().clone().to(dtype).requires_grad_(True)
        lp = F.log_softmax(lg, -1).transpose(0, 1)
        nll = F.ctc_loss(lp, tg, il, tl, blank=V - 1, reduction="none")
        F.ctc_loss(lp, tg, il, tl, blank=V - 1).backward()
        return nll.detach(), lg.grad
    n32, g32 = aten(torch.float32)
    n64, g64 = aten(torch.float64)
    ld = logits.to(DEV).requires_grad_(True)
    loss, nll = asr_amd.ctc_loss(ld, il.to(DEV), tg.to(DEV))
    loss.backward()
    grad, nll = ld.grad.cpu().double(), nll.cpu().double()
    print("repeats", repeats)
    print("  utt  T    U  | nll rel err: ours      aten32   | grad max abs err: ours      aten32    | max |g|   | err / (1e-5 + 2e-3 |g|): ours  aten32")
    for b in range(B):
        eo, ea = (grad[b] - g64[b]).abs(), (g32[b].double() - g64[b]).abs()
        den = 1e-5 + 2e-3 * g64[b].abs()
        print("  %3d %4d %3d | %.2e  %.2e | %.3e  %.3e | %.3e | %.2f  %.2f" % (
            b, int(il[b]), int(tl[b]), abs(float(nll[b] - n64[b])) / float(n64[b]), abs(float(n32[b].double() - n64[b])) / float(n64[b]),
            float(eo.max()), float(ea.max()), float(g64[b].abs().max()), float((eo / den).max()), float((ea / den).max())))
