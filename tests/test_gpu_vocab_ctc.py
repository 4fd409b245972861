"""The training step's CTC branch forward as two launches (csrc/vocab.hip: asr_vocab_proj_ctc - ctc_fc's projection writing fp16
logits, the rows' log-sum-exp AND the CTC table rows picked out of the fp32 accumulators on their way through LDS; csrc/ctc.hip:
asr_ctc_loss_fwd_table - the alpha / beta recursion on the finished table; transformer.py:119,148 + loss.py:41-43):
  * against torch fp32 on the same bf16 operands: logits (rounded once to fp16), row log-sum-exp (also with logits of magnitude 1500);
  * the table rows bit for bit against the streaming CTC forward's own pass over the f32 logits re-based on this launch's lse, nll and
    loss against that forward and against aten's F.log_softmax + F.ctc_loss on the CPU;
  * the gradient pass on the fp16 logits image against the one on f32 logits;
with blocks that straddle utterances, labels on chunk boundaries, repeated labels, empty and full-length targets, ragged lengths."""
import numpy as np
import pytest
import torch

from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()


def make(B, L, U, V, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(B * L, 256, generator=g).bfloat16()
    w = (torch.randn(V, 256, generator=g) * 0.15).bfloat16()
    tg = torch.randint(1, V - 1, (B, U), generator=g)
    if U > 6:
        tg[0, U - 2:] = 0                         # a shorter target
        tg[1 % B, 1] = tg[1 % B, 0]               # a repeated label
        tg[0, 0], tg[0, 1], tg[0, 2] = 63, 64, min(V - 2, 127)      # labels on the 64-column chunk boundaries
        tg[(B - 1), 3] = V - 2                    # the last label before the blank
        tg[(B - 1), 4] = 1
    if B > 2:
        tg[2, :] = 0                              # an empty target: the table row holds the blank alone
    il = torch.randint(max(2 * U + 2, L // 2), L + 1, (B,), generator=g)
    il[0] = L
    return x, w, tg, il


@pytest.mark.parametrize("B,L,U,V", [(3, 300, 51, 4234), (2, 128, 7, 130), (5, 257, 50, 1000), (4, 200, 63, 700), (2, 1000, 20, 4234), (2, 130, 3, 70), (1, 129, 1, 64)])
def test_vocab_proj_ctc_is_the_projection_plus_the_table_pass(B, L, U, V):
    x, w, tg, il = make(B, L, U, V, B * L + U)
    xd, wd, tgd, ild = x.to(DEV), w.to(DEV), tg.to(DEV), il.to(DEV)
    assert ops.vocab_proj_ctc_ok(xd, wd, B, L, U)
    logits16, loss, nll, st = ops.vocab_proj_ctc(xd, wd, tgd, ild, B, L)
    ref32 = x.float() @ w.float().t()                                        # fp32 accumulation of exact bf16 products
    assert logits16.dtype == torch.float16 and logits16.shape == (B * L, V) and logits16.stride(0) == (V + 7) // 8 * 8
    np.testing.assert_allclose(N(logits16), ref32.numpy(), atol=2e-3, rtol=1e-3)        # rounded once, to nearest even
    assert float((logits16.cpu() != ref32.half()).float().mean()) < 2e-3    # (the same value in all but the ties of another summation order)
    pad = torch.as_strided(logits16, (B * L, logits16.stride(0) - V), (logits16.stride(0), 1), logits16.storage_offset() + V)
    assert pad.numel() == 0 or float(pad.float().abs().max()) == 0.0        # the rows' pad columns: zeros, never -inf / NaN
    np.testing.assert_allclose(N(st.lse.view(-1)), torch.logsumexp(ref32, -1).numpy(), atol=2e-5, rtol=2e-6)
    # the streaming CTC forward on f32 logits (plain GEMM of the same operands): its table rows are (x - ITS lse) log2 e; re-based on
    # this launch's lse they must be this launch's rows, bit for bit in the gathered logit (the lse differ in the last bits)
    buf = torch.empty((B * L, (V + 7) // 8 * 8), device=DEV)
    ops.gemm_nt_raw(xd, B * L, 256, 256, wd, None, out=buf, ldc=buf.shape[1])
    logits32 = buf[:, :V]
    loss_t, nll_t, st_t = ops.ctc_loss_fwd(logits32.view(B, L, V), ild, tgd, n_chunks=1)
    L2E = 1.4426950408889634
    for b in range(B):
        n, sb = int(il[b]), 2 * int((tg[b] != 0).sum()) + 1
        mine, theirs = st.lp_ext[b, :n, :sb].double(), st_t.lp_ext[b, :n, :sb].double()
        d = (mine / L2E + st.lse[b, :n, None].double()) - (theirs / L2E + st_t.lse[b, :n, None].double())      # the gathered logits
        assert float(d.abs().max()) < 1e-5, b      # (two fp32 roundings of |x - lse| log2 e ~ 13 on either side)
        assert bool(torch.isinf(st.lp_ext[b, :n, sb:]).all()) and bool((st.lp_ext[b, :n, sb:] < 0).all())
    np.testing.assert_array_equal(N(st.tgt_len), (tg != 0).sum(1).numpy())
    np.testing.assert_allclose(N(nll), N(nll_t), rtol=2e-6)
    np.testing.assert_allclose(N(loss), N(loss_t), rtol=2e-6)
    st_s = st_t
    # aten on the CPU from the f32 logits
    lp = torch.log_softmax(logits32.float().cpu().double(), -1).view(B, L, V).transpose(0, 1)
    tl = (tg != 0).sum(1)
    ref = torch.nn.functional.ctc_loss(lp, tg, il, tl, blank=V - 1, reduction="none")
    np.testing.assert_allclose(N(nll), ref.numpy(), rtol=1e-5, atol=1e-4)
    # the gradient pass: fp16 logits image in, bf16 gradient image out - against the same pass on the f32 logits
    one = torch.ones(1, device=DEV)
    g16 = ops.ctc_loss_bwd(st, one, bf16=True)
    g32 = ops.ctc_loss_bwd(st_s, one, bf16=True)
    assert g16.dtype == torch.bfloat16 and g16.shape == (B, L, V)
    a, c = g16.float(), g32.float()
    # softmax of an fp16-rounded logit: exp(x +- ulp/2) = 0.2 % at |x| < 8 - under the bf16 image's own 0.4 % rounding: the two images
    # differ by at most one bf16 ulp of the larger gradient entries (a bf16 logits image measured 2.4 % there: 1.6 % at |x| ~ 5)
    assert float((a - c).norm() / c.norm()) < 4e-3
    assert float((a - c).abs().max()) <= 6e-3 * float(c.abs().max())
    for b in range(B):
        assert float(a[b, int(il[b]):].abs().max() if int(il[b]) < L else 0.0) == 0.0


def test_vocab_proj_ctc_north_star_shape_matches_aten():
    """(B 32, L 1000, U 51 incl. <eos>, V 4234): the S1 step's CTC branch; every utterance's nll against aten fp32 on the CPU."""
    B, L, U, V = 32, 1000, 51, 4234
    x, w, tg, il = make(B, L, U, V, 7)
    il[1] = L
    logits16, loss, nll, st = ops.vocab_proj_ctc(x.to(DEV), w.to(DEV), tg.to(DEV), il.to(DEV), B, L)
    ref_logits = x.float() @ w.float().t()
    lp = torch.log_softmax(ref_logits, -1).view(B, L, V).transpose(0, 1)
    tl = (tg != 0).sum(1)
    ref = torch.nn.functional.ctc_loss(lp, tg, il, tl, blank=V - 1, reduction="none")
    np.testing.assert_allclose(N(nll), ref.numpy(), rtol=1e-5, atol=1e-4)
    np.testing.assert_allclose(float(loss), float((ref / tl.clamp(min=1)).mean()), rtol=1e-5)
    np.testing.assert_allclose(N(logits16), ref_logits.numpy(), atol=2e-3, rtol=1e-3)      # fp16 image of fp32 sums of exact bf16 products


def test_vocab_proj_ctc_large_logits_do_not_overflow():
    g = torch.Generator().manual_seed(0)
    B, L, V = 1, 200, 300
    x = (torch.randn(B * L, 256, generator=g) * 6).bfloat16()
    w = (torch.randn(V, 256, generator=g) * 3).bfloat16()          # |logit| up to ~1500: exp() of the raw value overflows fp32
    tg = torch.randint(1, V - 1, (B, 5), generator=g)
    _, loss, nll, st = ops.vocab_proj_ctc(x.to(DEV), w.to(DEV), tg.to(DEV), torch.tensor([L]).to(DEV), B, L)
    ref = x.double() @ w.double().t()
    np.testing.assert_allclose(N(st.lse.view(-1)), torch.logsumexp(ref, -1).numpy(), rtol=2e-6, atol=1e-3)
    assert bool(torch.isfinite(st.lse).all()) and bool(torch.isfinite(nll).all())


def test_shapes_the_one_launch_table_does_not_take_fall_back():
    x = torch.randn(4 * 64, 256).bfloat16().to(DEV)
    w = torch.randn(70, 256).bfloat16().to(DEV)
    assert not ops.vocab_proj_ctc_ok(x, w, 4, 64, 7)            # L < 128: a block could span three utterances
    x = torch.randn(2 * 4096, 256).bfloat16().to(DEV)
    assert not ops.vocab_proj_ctc_ok(x, w, 2, 4096, 64)         # U + 1 > 64
    assert ops.vocab_proj_ctc_ok(x, w, 2, 4096, 63)


def test_trainer_step_on_the_table_form_equals_the_streaming_form(monkeypatch):
    """The whole CTC branch inside Trainer (side stream, projection + table, recursion, gradient pass on the fp16 image, ctc_fc's two
    backward GEMMs, the join into the encoder's gradient) against the same step with the branch on the plain GEMM + streaming CTC
    forward + f32-logits gradient pass: losses to 1e-5, ctc_fc's and the encoder's gradients to the bf16 images' rounding."""
    import asr_amd
    B, T, U, V = 8, 640, 20, 500
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, 80, generator=g).to(DEV)
    lens = torch.randint(T // 2, T + 1, (B,), generator=g)
    lens[0] = T
    tg = torch.randint(4, V - 1, (B, U), generator=g)
    tg[1, 12:] = 0
    grads, losses = [], []
    asr_amd.set_precision("bf16")
    for fused in (True, False):
        monkeypatch.setattr(ops, "FUSED_VOCAB_CTC", fused)
        torch.manual_seed(11)
        model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 4, 256, 512, dropout=0.0),
                                        asr_amd.Decoder(2, 3, V, 1, 4, 256, 512, dropout=0.0)).to(DEV).train()
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens.to(DEV), tg.to(DEV), max_target_len=U)
        if fused:
            assert tr._side is not None and tr._side["st"].logits.dtype == torch.float16      # the table form ran
        tr.backward(state)
        torch.cuda.synchronize()
        losses.append((float(ctc), float(ce)))
        grads.append({n: p.grad.detach().float().clone() for n, p in model.named_parameters()})
    np.testing.assert_allclose(losses[0], losses[1], rtol=1e-5)
    for name in ("ctc_fc.weight", "encoder.layer_stack.1.pos_ffn.w_2.weight", "encoder.layer_stack.0.slf_attn.w_qs.weight", "encoder.linear_in.weight"):
        a, c = grads[0][name], grads[1][name]
        assert float((a - c).norm() / c.norm()) < 1e-2, name


def test_launch_budget_changes_no_result():
    """ops.launch_budget (asr_launch_budget): the CTC gradient pass on 3 x 40 workgroups gives the bits of the full grid; the persistent
    data-gradient GEMM on 40 workgroups (split-K partial sums meet in atomics) and the weight-gradient GEMM (another split over the
    rows) the same sums to fp32 rounding."""
    B, L, U, V = 4, 300, 20, 1000
    x, w, tg, il = make(B, L, U, V, 5)
    xd, wd = x.to(DEV), w.to(DEV)
    _, _, _, st = ops.vocab_proj_ctc(xd, wd, tg.to(DEV), il.to(DEV), B, L)
    one = torch.ones(1, device=DEV)
    g_full = ops.ctc_loss_bwd(st, one, bf16=True).clone()
    g2 = torch.as_strided(g_full, (B * L, V), (g_full.stride(-2), 1), g_full.storage_offset())
    dx_full = ops.gemm_nn(g2, wd).clone()
    dw_full = ops.gemm_tn(g2, xd).clone()
    with ops.launch_budget(40):
        _, _, _, st_b = ops.vocab_proj_ctc(xd, wd, tg.to(DEV), il.to(DEV), B, L)
        g_b = ops.ctc_loss_bwd(st_b, one, bf16=True)
        dx_b = ops.gemm_nn(g2, wd)
        dw_b = ops.gemm_tn(g2, xd)
    assert int(ops.lib().asr_launch_budget(0)) == 0              # the context restored "none"
    assert torch.equal(g_b, g_full)
    np.testing.assert_allclose(N(dx_b), N(dx_full), rtol=1e-4, atol=1e-6)
    np.testing.assert_allclose(N(dw_b), N(dw_full), rtol=1e-4, atol=1e-6)
