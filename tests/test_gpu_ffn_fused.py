"""The one-launch feed-forward sub-layer (csrc/ffn.hip: asr_ffn_fwd / asr_ffn_bwd; module.py:48-53 + encoder.py:77) against a
torch-fp32 CPU evaluation of the reference's op sequence on the same bf16-rounded operands and the same dropout mask
(oracle.dropout_mask), at shapes with ragged lengths and a partial last block; then the module in train mode against the separate
launches it replaces and the whole encoder layer's gradients."""
import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops
from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()
THR = 6554   # p = 0.1


def _case(B, L, dff, seed):
    g = torch.Generator().manual_seed(seed)
    M = B * L
    x32 = torch.randn(M, 256, generator=g)
    w1 = (torch.randn(dff, 256, generator=g) * 0.06).bfloat16()
    w2 = (torch.randn(256, dff, generator=g) * 0.05).bfloat16()
    b1 = torch.randn(dff, generator=g) * 0.2
    b2 = torch.randn(256, generator=g) * 0.2
    gam = torch.rand(256, generator=g) + 0.5
    bet = torch.randn(256, generator=g) * 0.3
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    return x32, w1, b1, w2, b2, gam, bet, lens


def _reference(x32, w1, b1, w2, b2, gam, bet, lens, B, L, mask):
    """fp32 CPU: the reference's ops on the operands the kernel sees (x and the hidden activation rounded to bf16)."""
    x16 = x32.bfloat16().float()
    hid = torch.relu(x16 @ w1.float().t() + b1)
    hid16 = hid.bfloat16().float().requires_grad_(True)
    o = hid16 @ w2.float().t() + b2
    s = (o * mask if mask is not None else o) + x32
    y = torch.nn.functional.layer_norm(s, (256,), gam, bet, 1e-5)
    keep = (torch.arange(L)[None, :] < lens[:, None]).reshape(-1, 1).float()
    return hid16, s, y * keep


@pytest.mark.parametrize("B,L,dff,drop", [(4, 37, 128, False), (3, 100, 2048, True), (2, 128, 256, True), (1, 5, 64, False), (5, 129, 512, True)])
def test_ffn_fwd_against_torch_fp32(B, L, dff, drop):
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=B * 1000 + L)
    M = B * L
    mask = torch.from_numpy(O.dropout_mask((B, L, 256), THR, 5, 9)).view(M, 256) if drop else None
    hid_ref, s_ref, y_ref = _reference(x32, w1, b1, w2, b2, gam, bet, lens, B, L, mask)
    d = lambda t: t.to(DEV).contiguous()
    for train in (True, False):
        hid, bits, s, y32, y16, mean, rstd = ops.ffn_fwd(d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L,
                                                         row_len=d(lens).int(), train=train, drop_x=ops.Dropout(THR, 5, 9) if drop else None)
        np.testing.assert_allclose(N(y32), y_ref.detach().numpy(), atol=4e-3, rtol=2e-3)     # (fp32 sums of bf16 products in another order)
        np.testing.assert_allclose(N(y16), N(y32), atol=2e-2, rtol=8e-3)
        pad = (torch.arange(L)[None, :] >= lens[:, None]).reshape(-1)
        assert float(N(y32)[pad.numpy()].__abs__().max() if pad.any() else 0.0) == 0.0                   # padded rows: exact zeros (encoder.py:77)
        if train:
            np.testing.assert_allclose(N(hid), hid_ref.detach().numpy(), atol=2e-2, rtol=8e-3)        # one bf16 ulp
            np.testing.assert_allclose(N(s), s_ref.detach().numpy(), atol=4e-3, rtol=2e-3)
            mu = s_ref.detach().mean(-1)
            np.testing.assert_allclose(N(mean), mu.numpy(), atol=2e-4)
            np.testing.assert_allclose(N(rstd), (1.0 / torch.sqrt(s_ref.detach().var(-1, unbiased=False) + 1e-5)).numpy(), rtol=1e-3)
            if drop:        # the dropped positions are exactly the mask's (the pre-norm sum equals the residual there up to the kept part)
                np.testing.assert_array_equal(N(s)[(mask == 0).numpy()], x32.numpy()[(mask == 0).numpy()])
        else:
            assert hid is None and bits is None and s is None


@pytest.mark.parametrize("B,L,dff", [(4, 37, 128), (3, 100, 2048), (1, 5, 64), (5, 129, 512)])
def test_ffn_bwd_against_torch_autograd(B, L, dff):
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=7 * B + L)
    M = B * L
    g = torch.Generator().manual_seed(L)
    d = lambda t: t.to(DEV).contiguous()
    hid, bits, s, y32, y16, mean, rstd = ops.ffn_fwd(d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L,
                                                     row_len=d(lens).int(), train=True)
    ds32 = torch.randn(M, 256, generator=g) * 0.05
    ds16 = ds32.bfloat16()
    # reference: autograd through x -> relu(x W1^T + b1) -> . W2^T with the upstream gradient ds16 (bf16 operands as the kernel sees
    # them), masked by the KERNEL's own activation pattern (a unit whose pre-activation rounds across zero may differ by one ulp)
    act = N(hid) > 0
    dh = (ds16.float() @ w2.float()) * torch.from_numpy(act).float()
    dh16 = dh.bfloat16().float()
    dx = dh16 @ w1.float() + ds32
    d_hid, dxg = ops.ffn_bwd(d(ds16), d(ds32), d(w1), d(w2), bits)
    np.testing.assert_allclose(N(d_hid), dh16.numpy(), atol=2e-3, rtol=8e-3)
    assert int(((N(d_hid) != 0) & ~act).sum()) == 0                       # exact zeros where the unit was off
    np.testing.assert_allclose(N(dxg), dx.numpy(), atol=2e-3, rtol=2e-3)


def test_module_with_fused_ffn_equals_separate_launches(monkeypatch):
    """PositionwiseFeedForward forward + backward tape in train mode (dropout 0.1, ragged lengths): the fused sub-layer against the
    GEMM + GEMM + LayerNorm forward and the two data-gradient GEMMs it replaces, under the same dropout mask."""
    B, L = 3, 70
    torch.manual_seed(3)
    ffn = asr_amd.PositionwiseFeedForward(256, 512, dropout=0.1).to(DEV).train()
    x = torch.randn(B, L, 256, device=DEV)
    lens = torch.tensor([70, 61, 33], device=DEV)
    dy = torch.randn(B * L, 256, device=DEV) * 0.1
    from asr_amd import modules as Mo

    def run(rows):
        monkeypatch.setattr(ops, "FUSED_FFN_MIN_ROWS", rows)
        asr_amd.manual_seed(11)
        for p in ffn.parameters():
            p.grad = torch.zeros_like(p)
        with asr_amd.precision("bf16"), Mo.record() as tape:
            xa = Mo._act(x)
            xa.b16 = ops.cast_bf16(xa.f32)
            y = ffn._impl(xa, ops.as_i32(lens, x.device))
            y.grad = dy.clone()
            tape.backward()
        torch.cuda.synchronize()
        return N(y.f32), N(xa.grad), {n: N(p.grad) for n, p in ffn.named_parameters()}

    y0, dx0, g0 = run(1 << 30)       # separate launches
    y1, dx1, g1 = run(1)             # fused
    np.testing.assert_allclose(y1, y0, atol=3e-2, rtol=2e-2)
    np.testing.assert_array_equal(y1[70 + 61:70 + 70], 0.0)       # utterance 1's padded rows
    rel = lambda a, b: float(np.linalg.norm(a - b) / (np.linalg.norm(b) + 1e-12))
    assert rel(dx1, dx0) < 2e-2
    for n in g0:
        assert rel(g1[n], g0[n]) < 3e-2, n


@pytest.mark.parametrize("B,L,drop", [(32, 250, True), (5, 1000, False), (33, 129, True)])
def test_proj_ln_equals_the_gemm_and_layernorm_pair(B, L, drop, monkeypatch):
    """asr_proj_ln_fwd (attention.py:58-60 at encoder size: fc -> dropout -> + residual -> layer_norm, encoder.py:77 row mask) against
    the two launches it replaces - same operands, same dropout site, ragged lengths, a partial last block - and the saved statistics."""
    M = B * L
    g = torch.Generator().manual_seed(B + L)
    ctx = torch.randn(M, 256, generator=g).bfloat16().to(DEV)
    res = torch.randn(M, 256, generator=g).to(DEV)
    w = (torch.randn(256, 256, generator=g) * 0.06).bfloat16().to(DEV)
    bias = (torch.randn(256, generator=g) * 0.2).to(DEV)
    gam = (torch.rand(256, generator=g) + 0.5).to(DEV)
    bet = (torch.randn(256, generator=g) * 0.3).to(DEV)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    lens = lens.int().to(DEV)
    d = ops.Dropout(THR, 7, 3) if drop else None
    assert ops.proj_ln_ok(ctx, w, 256, B, L)
    s1, y32a, y16a, mean1, rstd1 = ops.proj_ln(ctx, w, bias, res, gam, bet, B, L, row_len=lens, save_stats=True, drop_x=d)
    o = ops.gemm_nt(ctx, w, bias)
    y32b, y16b, mean2, rstd2 = ops.add_layernorm(o, res, gam, bet, B, L, row_len=lens, want_bf16=True, save_stats=True, drop_x=d)
    np.testing.assert_allclose(N(s1), N(o), atol=2e-3, rtol=1e-3)          # (save mode leaves the pre-norm sum in o)
    np.testing.assert_allclose(N(y32a), N(y32b), atol=3e-3, rtol=1e-3)
    np.testing.assert_allclose(N(y16a), N(y16b), atol=2e-2, rtol=8e-3)
    np.testing.assert_allclose(N(mean1), N(mean2), atol=2e-4)
    np.testing.assert_allclose(N(rstd1), N(rstd2), rtol=1e-3)
    pad = (torch.arange(L, device=DEV)[None, :] >= lens[:, None]).reshape(-1)
    assert float(y32a[pad].abs().max() if pad.any() else 0.0) == 0.0
    if drop:        # the dropped positions are the same ones: there the pre-norm sum is the residual itself
        m = torch.from_numpy(O.dropout_mask((B, L, 256), THR, 7, 3)).view(M, 256).to(DEV)
        assert torch.equal(s1[m == 0], res[m == 0])
    s0, y32c, _, _, _ = ops.proj_ln(ctx, w, bias, res, gam, bet, B, L, row_len=lens, save_stats=False, drop_x=d)      # eval form
    assert s0 is None and torch.equal(y32c, y32a)


@pytest.mark.parametrize("B,L,n_proj,h,scale", [(32, 1000, 3, 4, 0.125 * 1.4426950408889634), (32, 1000, 12, 4, 1.0), (33, 517, 1, 4, 1.0),
                                                (17, 1001, 3, 1, 0.5), (2, 9000, 2, 4, 1.0)])
def test_proj_heads_rows_kernel_equals_the_tiled_gemm(B, L, n_proj, h, scale, monkeypatch):
    """asr_proj_heads at encoder size (ffn.hip: proj_heads_rows_kernel; attention.py:43-49) against the tiled GEMM it replaces and
    against torch fp32 on the bf16-rounded operands: all projections / heads, the Q scale, a ragged last row block, odd chunk counts."""
    g = torch.Generator().manual_seed(B * L + n_proj)
    M, Nn = B * L, n_proj * h * 64
    x = torch.randn(M, 256, generator=g).bfloat16()
    w = (torch.randn(Nn, 256, generator=g) * 0.06).bfloat16()
    bias = torch.randn(Nn, generator=g) * 0.1
    xd, wd, bd = x.to(DEV), w.to(DEV), bias.to(DEV)
    monkeypatch.setenv("ASR_AMD_HEADS_ROWS", "0")
    old = ops.proj_heads(xd, wd, bd, n_proj, B, L, h, scale)
    monkeypatch.setenv("ASR_AMD_HEADS_ROWS", "1")
    new = torch.full((n_proj, B, h, L, 64), float("nan"), device=DEV, dtype=torch.bfloat16)
    new.copy_(ops.proj_heads(xd, wd, bd, n_proj, B, L, h, scale))
    torch.cuda.synchronize()
    ref = (x.float() @ w.float().t() + bias).view(B, L, n_proj, h, 64).permute(2, 0, 3, 1, 4).contiguous()
    ref[0] *= scale
    assert torch.isfinite(new.float()).all()
    # against fp32: one bf16 rounding of a value of magnitude ~1
    np.testing.assert_allclose(new.float().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=1e-2)
    # against the tiled GEMM: the same products summed in another order - at most one bf16 ulp apart, and almost everywhere equal
    d = (new.float() - old.float()).abs().cpu()
    assert float(d.max()) <= 2.0 ** -7 * max(1.0, float(ref.abs().max()))
    assert float((d > 0).float().mean()) < 0.02


@pytest.mark.parametrize("B,L,dff,drop", [(4, 37, 128, False), (3, 100, 2048, True), (1, 5, 64, True), (5, 129, 512, True)])
def test_ffn_bwd_with_the_layernorm_backward_folded_in(B, L, dff, drop):
    """asr_ffn_bwd_ln = asr_ffn_bwd followed by asr_add_layernorm_bwd on its dx (the LayerNorm that produced the sub-layer's input,
    attention.py:60): the same ds / ds16 / column sums, without dx in memory.  Same arithmetic per row, so the f32 outputs agree to the
    last bits (the column sums to float-atomic order); ragged lengths, a partial last block, dropout on the projection's gradient."""
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=11 * B + L)
    M = B * L
    g = torch.Generator().manual_seed(L + 1)
    d = lambda t: t.to(DEV).contiguous()
    hid, bits, s, y32, y16, mean, rstd = ops.ffn_fwd(d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L,
                                                     row_len=d(lens).int(), train=True)
    ds32 = d(torch.randn(M, 256, generator=g) * 0.05)
    ds16 = ds32.bfloat16()
    # the LayerNorm in front of the sub-layer: its saved pre-norm sum / statistics, its own parameters and row mask
    p_s = d(torch.randn(M, 256, generator=g))
    p_mean, p_var = p_s.mean(-1), p_s.var(-1, unbiased=False)
    p_rstd = 1.0 / torch.sqrt(p_var + 1e-5)
    p_gam = d(torch.rand(256, generator=g) + 0.5)
    lens_d = d(lens).int()
    dp = ops.Dropout(THR, 3, 4) if drop else None
    dg0, db0, dbias0 = (torch.zeros(256, device=DEV) for _ in range(3))
    d_hid0, dx = ops.ffn_bwd(ds16, ds32, d(w1), d(w2), bits)
    ds_ref, ds16_ref = ops.add_layernorm_bwd(dx, p_s, p_mean, p_rstd, p_gam, lens_d, B, L, dg0, db0, want_bf16=True, dbias=dbias0, drop_x=dp)
    dg1, db1, dbias1 = (torch.zeros(256, device=DEV) for _ in range(3))
    d_hid1, ds_f, ds16_f = ops.ffn_bwd_ln(ds16, ds32, d(w1), d(w2), bits, B, L, p_s, p_mean, p_rstd, p_gam, lens_d, dg1, db1, dbias=dbias1,
                                          drop_x=dp)
    np.testing.assert_array_equal(N(d_hid1), N(d_hid0))
    np.testing.assert_allclose(N(ds_f), N(ds_ref), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(N(ds16_f), N(ds16_ref), atol=1e-6, rtol=1e-2)      # (a bf16 ulp where the f32 value sits on a rounding boundary)
    pad = (torch.arange(L)[None, :] >= lens[:, None]).reshape(-1).numpy()
    if pad.any():
        assert float(np.abs(N(ds_f)[pad]).max()) == 0.0                           # masked rows: no gradient
    for a, b in ((dg1, dg0), (db1, db0), (dbias1, dbias0)):
        np.testing.assert_allclose(N(a), N(b), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,L,dff,drop", [(4, 37, 128, False), (3, 100, 2048, True), (2, 128, 256, True), (1, 5, 64, False), (5, 129, 512, True),
                                           (32, 250, 2048, True)])
def test_attn_ffn_fwd_is_the_two_launches(B, L, dff, drop):
    """asr_attn_ffn_fwd (the attention sub-layer's tail in front of the feed-forward sub-layer, one launch) against asr_proj_ln_fwd followed by
    asr_ffn_fwd: every output tensor bit for bit, training and inference form, ragged lengths, partial last block."""
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=B * 77 + L)
    g = torch.Generator().manual_seed(B + L + dff)
    M = B * L
    ctx = torch.randn(M, 256, generator=g).bfloat16()
    wo = (torch.randn(256, 256, generator=g) * 0.06).bfloat16()
    bo = torch.randn(256, generator=g) * 0.2
    g0 = torch.rand(256, generator=g) + 0.5
    be0 = torch.randn(256, generator=g) * 0.3
    d = lambda t: t.to(DEV).contiguous()
    ctx, wo, bo, g0, be0, x32, w1, b1, w2, b2, gam, bet = [d(t) for t in (ctx, wo, bo, g0, be0, x32, w1, b1, w2, b2, gam, bet)]
    rl = d(lens).int()
    dr0 = ops.Dropout(THR, 11, 3) if drop else None
    dr1 = ops.Dropout(THR, 5, 9) if drop else None
    for train in (True, False):
        for save_s in ((True, False) if train else (True,)):
            s0, y0_32, y0_16, mean0, rstd0 = ops.proj_ln(ctx, wo, bo, x32, g0, be0, B, L, row_len=rl, save_stats=train, drop_x=dr0, save_s=save_s)
            ref = ops.ffn_fwd(y0_16, y0_32, w1, b1, w2, b2, gam, bet, B, L, row_len=rl, train=train, drop_x=dr1, save_s=save_s)
            pre, main = ops.attn_ffn_fwd(ctx, wo, bo, x32, g0, be0, w1, b1, w2, b2, gam, bet, B, L, row_len=rl, train=train, drop0=dr0, drop_x=dr1,
                                         save_s=save_s)
            for name, a_, b_ in list(zip(("s0", "x32", "x16", "mean0", "rstd0"), pre, (s0, y0_32, y0_16, mean0, rstd0))) + \
                    list(zip(("hid", "bits", "s", "y32", "y16", "mean", "rstd"), main, ref)):
                assert (a_ is None) == (b_ is None), name
                if a_ is not None:
                    if name == "bits":      # (words of rows past M are never written)
                        continue
                    assert torch.equal(a_, b_), "%s differs (train %s, save_s %s): max %g" % (name, train, save_s, float((a_.float() - b_.float()).abs().max()))
