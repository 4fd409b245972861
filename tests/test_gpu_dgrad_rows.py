"""The row-block data gradient of a d_model-input nn.Linear (csrc/dgrad_rows.hip: asr_dgrad_rows / asr_dgrad_rows_ln) against a torch
fp32 evaluation on the same bf16 operands, and its LayerNorm-backward epilogue against asr_add_layernorm_bwd on the separate GEMM's
result (autograd of attention.py:43-49, :58-60 and module.py:52 at encoder size)."""
import numpy as np
import pytest
import torch

from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()
THR = 6554   # p = 0.1


def _case(M, K, seed, wide=0):
    g = torch.Generator().manual_seed(seed)
    dy = (torch.randn(M, K + wide, generator=g) * 0.1).bfloat16()
    w = (torch.randn(K, 256, generator=g) * 0.05).bfloat16()
    add = torch.randn(M, 256, generator=g) * 0.1
    return dy, w, add


@pytest.mark.parametrize("M,K,wide", [(300, 64, 0), (1000, 768, 0), (517, 256, 512), (129, 3072, 0)])
def test_dgrad_rows_against_torch(M, K, wide, monkeypatch):
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN", 1)
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN_K", 64)
    dy, w, add = _case(M, K, M + K, wide)
    a = dy.to(DEV)[:, wide // 2:wide // 2 + K] if wide else dy.to(DEV)          # a column slice of a wider buffer (dqkv, dkv)
    ref = a.float().cpu() @ w.float()
    assert ops.dgrad_rows_ok(a, w.to(DEV))
    out = ops.gemm_nn(a, w.to(DEV))
    np.testing.assert_allclose(N(out), ref.numpy(), atol=2e-4 * K ** 0.5, rtol=2e-3)
    out2 = ops.gemm_nn(a, w.to(DEV), addend=add.to(DEV))
    np.testing.assert_allclose(N(out2), (ref + add).numpy(), atol=2e-4 * K ** 0.5, rtol=2e-3)
    out16 = ops.gemm_nn(a, w.to(DEV), out_dtype=torch.bfloat16)
    np.testing.assert_allclose(N(out16), ref.numpy(), atol=2e-4 * K ** 0.5 + 4e-3, rtol=1e-2)
    # ... and the tiled GEMM it replaces at these shapes computes the same thing
    monkeypatch.setattr(ops, "DGRAD_ROWS", False)
    np.testing.assert_allclose(N(out2), N(ops.gemm_nn(a, w.to(DEV), addend=add.to(DEV))), atol=1e-4 * K ** 0.5, rtol=1e-3)


@pytest.mark.parametrize("B,L,K,drop", [(3, 100, 768, True), (5, 129, 256, False), (2, 64, 64, True)])
def test_dgrad_rows_with_the_layernorm_backward_folded_in(B, L, K, drop, monkeypatch):
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN", 1)
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN_K", 64)
    M = B * L
    dy, w, add = _case(M, K, 3 * M + K)
    g = torch.Generator().manual_seed(K)
    d = lambda t: t.to(DEV).contiguous()
    p_s = d(torch.randn(M, 256, generator=g))
    p_mean, p_rstd = p_s.mean(-1), 1.0 / torch.sqrt(p_s.var(-1, unbiased=False) + 1e-5)
    p_gam = d(torch.rand(256, generator=g) + 0.5)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    lens_d = d(lens).int()
    dp = ops.Dropout(THR, 3, 4) if drop else None
    dg0, db0, dbias0 = (torch.zeros(256, device=DEV) for _ in range(3))
    dx = ops.gemm_nn(d(dy), d(w), addend=d(add))
    ds_ref, ds16_ref = ops.add_layernorm_bwd(dx, p_s, p_mean, p_rstd, p_gam, lens_d, B, L, dg0, db0, want_bf16=True, dbias=dbias0, drop_x=dp)
    dg1, db1, dbias1 = (torch.zeros(256, device=DEV) for _ in range(3))
    ds_f, ds16_f = ops.gemm_nn_ln(d(dy), d(w), d(add), B, L, p_s, p_mean, p_rstd, p_gam, lens_d, dg1, db1, dbias=dbias1, drop_x=dp)
    np.testing.assert_allclose(N(ds_f), N(ds_ref), atol=1e-6, rtol=1e-5)
    np.testing.assert_allclose(N(ds16_f), N(ds16_ref), atol=1e-6, rtol=1e-2)
    pad = (torch.arange(L)[None, :] >= lens[:, None]).reshape(-1).numpy()
    if pad.any():
        assert float(np.abs(N(ds_f)[pad]).max()) == 0.0
    for x, y in ((dg1, dg0), (db1, db0), (dbias1, dbias0)):
        np.testing.assert_allclose(N(x), N(y), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,L,drop", [(3, 100, True), (2, 257, False)])
def test_layernorm_backward_from_the_output_instead_of_the_prenorm_sum(B, L, drop, monkeypatch):
    """A forward that kept no pre-norm sum (ASR_AMD_LN_FROM_Y: asr_ffn_fwd / asr_proj_ln_fwd with s_out = NULL) hands the backward the
    LayerNorm's output: x^ = (y - beta) / gamma.  Same gradients as from (s, mean, rstd) - stand-alone kernel and both folded forms."""
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN", 1)
    monkeypatch.setattr(ops, "DGRAD_ROWS_MIN_K", 64)
    M, K = B * L, 128
    g = torch.Generator().manual_seed(B * L)
    d = lambda t: t.to(DEV).contiguous()
    s = d(torch.randn(M, 256, generator=g) * 2.0 + 0.3)
    mean, rstd = s.mean(-1), 1.0 / torch.sqrt(s.var(-1, unbiased=False) + 1e-5)
    gam, bet = d(torch.rand(256, generator=g) + 0.5), d(torch.randn(256, generator=g) * 0.3)
    lens = torch.randint(max(1, L // 2), L + 1, (B,), generator=g)
    lens[0] = L
    keep = d((torch.arange(L)[None, :] < lens[:, None]).reshape(-1, 1).float())
    y = ((s - mean[:, None]) * rstd[:, None] * gam + bet) * keep          # what the forward leaves (masked rows zeroed, encoder.py:77)
    dyv = d(torch.randn(M, 256, generator=g) * 0.1)
    lens_d = d(lens).int()
    dp = ops.Dropout(THR, 3, 4) if drop else None

    def run(fn):
        acc = [torch.zeros(256, device=DEV) for _ in range(3)]
        out = fn(acc)
        return [N(t) for t in out] + [N(t) for t in acc]
    ref = run(lambda a: ops.add_layernorm_bwd(dyv, s, mean, rstd, gam, lens_d, B, L, a[0], a[1], want_bf16=True, dbias=a[2], drop_x=dp))
    got = run(lambda a: ops.add_layernorm_bwd(dyv, y, None, rstd, gam, lens_d, B, L, a[0], a[1], want_bf16=True, dbias=a[2], drop_x=dp, beta=bet))
    for x, r, tol in zip(got, ref, (2e-6, 1e-2, 5e-4, 5e-4, 5e-4)):
        np.testing.assert_allclose(x, r, atol=tol, rtol=1e-4 if tol < 1e-3 else 1e-2)
    # the folded form of the data-gradient launch, both conventions
    dy16, w, add = _case(M, K, 5)
    ref = run(lambda a: ops.gemm_nn_ln(d(dy16), d(w), d(add), B, L, s, mean, rstd, gam, lens_d, a[0], a[1], dbias=a[2], drop_x=dp))
    got = run(lambda a: ops.gemm_nn_ln(d(dy16), d(w), d(add), B, L, y, None, rstd, gam, lens_d, a[0], a[1], dbias=a[2], drop_x=dp, ln_beta=bet))
    for x, r, tol in zip(got, ref, (2e-6, 1e-2, 5e-4, 5e-4, 5e-4)):
        np.testing.assert_allclose(x, r, atol=tol, rtol=1e-4 if tol < 1e-3 else 1e-2)
