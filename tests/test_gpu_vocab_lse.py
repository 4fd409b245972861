"""The CTC branch's vocabulary projection with the row log-sum-exp from the same launch (csrc/vocab.hip: asr_vocab_proj_lse;
transformer.py:119,148 + loss.py:41) against torch fp32 on the same bf16 operands, and the CTC forward that starts from that lse
(asr_ctc_loss_fwd_lse: label gather + recursion) against the form that streams the logits itself and against F.ctc_loss."""
import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()


@pytest.mark.parametrize("M,V", [(300, 4234), (128, 64), (37, 130), (1, 70), (513, 1000)])
def test_vocab_proj_lse_against_torch_fp32(M, V):
    g = torch.Generator().manual_seed(M + V)
    x = torch.randn(M, 256, generator=g).bfloat16()
    w = (torch.randn(V, 256, generator=g) * 0.2).bfloat16()
    logits, lse = ops.vocab_proj_lse(x.to(DEV), w.to(DEV))
    ref = x.float() @ w.float().t()
    assert logits.shape == (M, V) and logits.stride(0) == (V + 7) // 8 * 8
    np.testing.assert_allclose(N(logits), ref.numpy(), atol=2e-4, rtol=2e-5)          # fp32 accumulation of exact bf16 products
    np.testing.assert_allclose(N(lse), torch.logsumexp(ref, -1).numpy(), atol=2e-5, rtol=2e-6)
    pad = torch.as_strided(logits, (M, logits.stride(0) - V), (logits.stride(0), 1), logits.storage_offset() + V)
    assert float(pad.abs().max()) == 0.0 if pad.numel() else True                       # the rows' pad columns: zeros, never -inf / NaN


def test_vocab_proj_lse_large_logits_do_not_overflow():
    g = torch.Generator().manual_seed(0)
    x = (torch.randn(200, 256, generator=g) * 6).bfloat16()
    w = (torch.randn(300, 256, generator=g) * 3).bfloat16()          # |logit| up to ~1500: exp() of the raw value overflows fp32
    logits, lse = ops.vocab_proj_lse(x.to(DEV), w.to(DEV))
    ref = x.double() @ w.double().t()
    np.testing.assert_allclose(N(lse), torch.logsumexp(ref, -1).numpy(), rtol=2e-6, atol=1e-3)
    assert bool(torch.isfinite(lse).all())


@pytest.mark.parametrize("B,L,U,V", [(3, 100, 7, 130), (2, 257, 50, 4234), (4, 64, 1, 70)])
def test_ctc_forward_from_row_lse(B, L, U, V):
    g = torch.Generator().manual_seed(B * L + U)
    x = torch.randn(B * L, 256, generator=g).bfloat16()
    w = (torch.randn(V, 256, generator=g) * 0.15).bfloat16()
    tg = torch.randint(1, V - 1, (B, U), generator=g)
    if U > 3:
        tg[0, U - 2:] = 0            # a shorter target
        tg[1, 1] = tg[1, 0]          # a repeated label
    il = torch.randint(max(2 * U + 2, L // 2), L + 1, (B,), generator=g)
    il[0] = L
    logits, lse = ops.vocab_proj_lse(x.to(DEV), w.to(DEV))
    l3 = logits.view(B, L, V)
    loss_a, nll_a, st_a = ops.ctc_loss_fwd(l3, il.to(DEV), tg.to(DEV), lse=lse)
    loss_b, nll_b, st_b = ops.ctc_loss_fwd(l3, il.to(DEV), tg.to(DEV), n_chunks=1)
    np.testing.assert_allclose(N(nll_a), N(nll_b), rtol=2e-6)
    np.testing.assert_allclose(N(loss_a), N(loss_b), rtol=2e-6)
    # aten on the CPU (the oracle configs[2] names), from the same logits
    lp = torch.log_softmax(logits.float().cpu().double(), -1).view(B, L, V).transpose(0, 1)
    tl = (tg != 0).sum(1)
    ref = torch.nn.functional.ctc_loss(lp, tg, il, tl, blank=V - 1, reduction="none")
    np.testing.assert_allclose(N(nll_a), ref.numpy(), rtol=1e-5, atol=1e-4)
    # the backward runs from either forward's state: same gradient
    ga = ops.ctc_loss_bwd(st_a, torch.ones(1, device=DEV))
    gb = ops.ctc_loss_bwd(st_b, torch.ones(1, device=DEV))
    np.testing.assert_allclose(N(ga), N(gb), atol=5e-6, rtol=1e-3)          # (the two row lse agree to the last bits only)
