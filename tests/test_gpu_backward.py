"""GPU parity of the backward-pass kernels against torch autograd on CPU (fp32)."""
import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops

pytestmark = pytest.mark.gpu
LOG2E = 1.4426950408889634
DEV = "cuda:0"


def N(t):
    return t.detach().float().cpu().numpy()


@pytest.mark.parametrize("M,N_,K", [(1000, 256, 2048), (777, 4236, 64), (300, 768, 256), (64, 128, 128), (1024, 256, 2048), (4096, 768, 256),
                                    (3200, 2048, 256), (128, 128, 256), (1632, 256, 256), (1000, 2048, 256), (72, 128, 128)])
@pytest.mark.parametrize("dts", [("bf16", "bf16"), ("f32", "bf16"), ("f32", "f32")])
def test_gemm_tn(M, N_, K, dts):
    g = torch.Generator().manual_seed(M + K)
    a = torch.randn(M, N_, generator=g)
    b = torch.randn(M, K, generator=g)
    ad = a.to(DEV).to(torch.bfloat16 if dts[0] == "bf16" else torch.float32)
    bd = b.to(DEV).to(torch.bfloat16 if dts[1] == "bf16" else torch.float32)
    cs = torch.zeros(N_, device=DEV)
    out = ops.gemm_tn(ad, bd, colsum=cs)
    ref = a.bfloat16().float().t() @ b.bfloat16().float()
    np.testing.assert_allclose(N(cs), (a.bfloat16().float() if dts[0] == "bf16" else a).sum(0).numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    np.testing.assert_allclose(N(out), ref.numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)
    out2 = ops.gemm_tn(ad, bd, out=out, accumulate=True)
    np.testing.assert_allclose(N(out2), 2 * ref.numpy(), atol=4e-2 * (M / 300) ** 0.5, rtol=2e-3)


@pytest.mark.parametrize("M,N_,K", [(1000, 256, 2048), (300, 80, 256), (77, 256, 768), (500, 256, 4240), (1000, 2048, 256), (130, 128, 64)])
def test_gemm_nn(M, N_, K):
    g = torch.Generator().manual_seed(M + K)
    dy = torch.randn(M, K, generator=g)
    w = torch.randn(K, N_, generator=g) / K ** 0.5
    add = torch.randn(M, N_, generator=g)
    out = ops.gemm_nn(dy.to(DEV), w.to(DEV).bfloat16(), addend=add.to(DEV))
    ref = dy.bfloat16().float() @ w.bfloat16().float() + add
    np.testing.assert_allclose(N(out), ref.numpy(), atol=3e-3, rtol=2e-3)
    out16 = ops.gemm_nn(dy.to(DEV).bfloat16(), w.to(DEV).bfloat16(), out_dtype=torch.bfloat16)
    np.testing.assert_allclose(N(out16), (ref - add).numpy(), atol=3e-2, rtol=2e-2)


@pytest.mark.parametrize("M,N_,K", [(1632, 256, 2048), (1632, 256, 768), (200, 128, 1024), (1632, 256, 3072), (640, 384, 512)])
def test_split_k_gemms(M, N_, K):
    """Few output tiles + long K + plain f32 C: the persistent NT / NN kernels cut K across workgroups and add partial tiles with
    float atomics into a zeroed C (gemm.hip: pick_ksplit); bias / addend are applied by the first split only."""
    g = torch.Generator().manual_seed(M + N_ + K)
    a = torch.randn(M, K, generator=g).bfloat16()
    w_nt = (torch.randn(N_, K, generator=g) / K ** 0.5).bfloat16()
    bias = torch.randn(N_, generator=g)
    add = torch.randn(M, N_, generator=g)
    ref_nt = a.float() @ w_nt.float().t() + bias
    for _ in range(2):      # twice: the result must not depend on what an earlier launch left in a recycled C buffer
        out = ops.gemm_nt(a.to(DEV), w_nt.to(DEV), bias.to(DEV))
        np.testing.assert_allclose(N(out), ref_nt.numpy(), atol=3e-3, rtol=2e-3)
    w_nn = (torch.randn(K, N_, generator=g) / K ** 0.5).bfloat16()
    ref_nn = a.float() @ w_nn.float() + add
    for _ in range(2):
        out = ops.gemm_nn(a.to(DEV), w_nn.to(DEV), addend=add.to(DEV))
        np.testing.assert_allclose(N(out), ref_nn.numpy(), atol=3e-3, rtol=2e-3)
    np.testing.assert_allclose(N(ops.gemm_nn(a.to(DEV), w_nn.to(DEV))), (ref_nn - add).numpy(), atol=3e-3, rtol=2e-3)


@pytest.mark.parametrize("M,V", [(2048, 4234), (1000, 131), (1632, 4234)])
def test_vocab_backward_gemms_on_a_padded_bf16_gradient(M, V):
    """The projection's backward on the bf16 gradient image (rows zero-padded to a multiple of 128): the LDS-DMA NN kernel with a K
    that is no multiple of 64 (weight rows past K-1 clamped) and the LDS-DMA TN kernel with an N that is no multiple of 128."""
    g = torch.Generator().manual_seed(M + V)
    D = 256
    Vp = (V + 127) // 128 * 128
    buf = torch.zeros(M, Vp)
    buf[:, :V] = torch.randn(M, V, generator=g) * 0.1
    gd = buf.bfloat16().to(DEV)
    g2 = gd[:, :V]                                     # [M, V] view, row stride Vp
    w = (torch.randn(V, D, generator=g) / V ** 0.5).bfloat16()
    x = torch.randn(M, D, generator=g).bfloat16()
    add = torch.randn(M, D, generator=g)
    dx = ops.gemm_nn(g2, w.to(DEV), addend=add.to(DEV))
    ref_dx = buf[:, :V].bfloat16().float() @ w.float() + add
    np.testing.assert_allclose(N(dx), ref_dx.numpy(), atol=3e-3, rtol=2e-3)
    dw = torch.zeros(V, D, device=DEV)
    ops.gemm_tn(g2, x.to(DEV), out=dw, accumulate=True)
    ref_dw = buf[:, :V].bfloat16().float().t() @ x.float()
    np.testing.assert_allclose(N(dw), ref_dw.numpy(), atol=2e-2 * (M / 300) ** 0.5, rtol=2e-3)


def test_colsum_and_embed_bwd():
    g = torch.Generator().manual_seed(0)
    a = torch.randn(1000, 300, generator=g)
    np.testing.assert_allclose(N(ops.colsum(a.to(DEV))), a.sum(0).numpy(), atol=1e-3)
    np.testing.assert_allclose(N(ops.colsum(a.to(DEV).bfloat16())), a.bfloat16().float().sum(0).numpy(), atol=1e-3)
    big = torch.randn(5000, 768, generator=g)
    np.testing.assert_allclose(N(ops.colsum(big.to(DEV).bfloat16())), big.bfloat16().float().sum(0).numpy(), atol=5e-3)
    odd = torch.randn(333, 50, generator=g)          # N not a multiple of 4, lda not a multiple of 4
    np.testing.assert_allclose(N(ops.colsum(odd.to(DEV))), odd.sum(0).numpy(), atol=1e-3)
    ids = torch.randint(0, 50, (200,), generator=g)
    dy = torch.randn(200, 64, generator=g)
    demb = torch.zeros(50, 64, device=DEV)
    ops.embed_bwd(ids.to(DEV), dy.to(DEV), demb)
    ref = torch.zeros(50, 64).index_add_(0, ids, dy)
    np.testing.assert_allclose(N(demb), ref.numpy(), atol=1e-4)


def test_gemm_nt_ex_addend_and_relu_mask():
    g = torch.Generator().manual_seed(1)
    a, w = torch.randn(300, 128, generator=g), torch.randn(200, 128, generator=g) / 11
    add = torch.randn(300, 200, generator=g)
    mask = torch.relu(torch.randn(300, 200, generator=g))
    out = ops.gemm_nt_ex(a.to(DEV), w.to(DEV), addend=add.to(DEV), relu_mask=mask.to(DEV).bfloat16())
    ref = (a @ w.t() + add) * (mask.bfloat16().float() > 0)
    np.testing.assert_allclose(N(out), ref.numpy(), atol=1e-4, rtol=1e-5)


@pytest.mark.parametrize("D", [64, 256, 512])
def test_add_layernorm_bwd(D):
    B, L = 3, 50
    g = torch.Generator().manual_seed(D)
    x = torch.randn(B * L, D, generator=g, requires_grad=True)
    r = torch.randn(B * L, D, generator=g)
    gam = torch.randn(D, generator=g, requires_grad=True)
    bet = torch.randn(D, generator=g, requires_grad=True)
    lens = torch.tensor([50, 31, 1], dtype=torch.int32)
    dy = torch.randn(B * L, D, generator=g)
    keep = (torch.arange(L)[None, :] < lens[:, None]).reshape(-1, 1).float()
    y = torch.nn.functional.layer_norm(x + r, (D,), gam, bet) * keep
    y.backward(dy)
    xd = x.detach().to(DEV).clone()
    y32, _, mean, rstd = ops.add_layernorm(xd, r.to(DEV), gam.detach().to(DEV), bet.detach().to(DEV), B, L, row_len=lens.to(DEV),
                                           save_stats=True)
    np.testing.assert_allclose(N(xd), (x + r).detach().numpy(), atol=1e-6)   # pre-norm sum saved in place
    dgam, dbet = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    dbias = torch.zeros(D, device=DEV)
    ds, ds16 = ops.add_layernorm_bwd(dy.to(DEV), xd, mean, rstd, gam.detach().to(DEV), lens.to(DEV), B, L, dgam, dbet, want_bf16=True,
                                     dbias=dbias)
    np.testing.assert_allclose(N(ds), x.grad.numpy(), atol=2e-5, rtol=1e-4)
    np.testing.assert_allclose(N(dbias), x.grad.sum(0).numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(N(dgam), gam.grad.numpy(), atol=2e-4, rtol=1e-4)
    np.testing.assert_allclose(N(dbet), bet.grad.numpy(), atol=2e-4, rtol=1e-4)


@pytest.mark.parametrize("B,h,Lq,Lk,causal,ragged", [(2, 2, 25, 25, False, True), (2, 4, 200, 200, False, True), (2, 2, 51, 51, True, True),
                                                     (2, 2, 51, 250, False, True), (1, 2, 300, 300, True, False), (1, 1, 1000, 1000, False, True),
                                                     (2, 2, 51, 1000, False, True), (2, 2, 40, 300, False, True), (1, 2, 64, 700, False, False)])   # 2 / 4 key streams
def test_attention_bwd(B, h, Lq, Lk, causal, ragged):
    g = torch.Generator().manual_seed(Lq * 3 + Lk)
    # device q carries log2(e) (asr_hip.h); the reference differentiates wrt exactly that tensor / log2(e)
    qdev = (torch.randn(B, h, Lq, 64, generator=g) * 0.4 * LOG2E).bfloat16()
    q = (qdev.float() / LOG2E).requires_grad_(True)
    k = torch.randn(B, h, Lk, 64, generator=g).bfloat16().float().requires_grad_(True)
    v = torch.randn(B, h, Lk, 64, generator=g).bfloat16().float().requires_grad_(True)
    k_len = None
    if ragged:
        k_len = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        k_len[0] = Lk
    s = q @ k.transpose(-1, -2)
    mask = torch.zeros(B, 1, Lq, Lk, dtype=torch.bool)
    if k_len is not None:
        mask |= (torch.arange(Lk)[None, :] >= k_len[:, None])[:, None, None, :]
    if causal:
        mask |= torch.triu(torch.ones(Lq, Lk, dtype=torch.bool), 1)[None, None]
    p = torch.softmax(s.masked_fill(mask, float("-inf")), -1)
    ctx = (p @ v).permute(0, 2, 1, 3).reshape(B, Lq, h * 64)
    dctx = torch.randn(B, Lq, h * 64, generator=g).bfloat16().float()
    ctx.backward(dctx)
    qd, kd, vd = qdev.to(DEV), k.detach().to(DEV).bfloat16(), v.detach().to(DEV).bfloat16()
    kl = None if k_len is None else k_len.to(DEV).int()
    ctx_d, lse = ops.attention_fwd(qd, kd, vd, kl, causal, need_lse=True)
    dq = torch.zeros(B * Lq, h * 64, device=DEV, dtype=torch.bfloat16)
    dkv = torch.zeros(B * Lk, 2 * h * 64, device=DEV, dtype=torch.bfloat16)
    ops.attention_bwd(qd, kd, vd, ctx_d, dctx.to(DEV).bfloat16(), lse, kl, causal, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:])
    to_tok = lambda t: t.permute(0, 2, 1, 3).reshape(t.shape[0] * t.shape[2], h * 64)
    # bf16 operands (P, dS, dO rounded to bf16 before their products) and bf16 outputs, fp32 accumulation.  A parity bound must not
    # grow with what the kernel happens to need, so it is stated relative to the gradient itself (see attn_bwd_errors): per tensor the
    # relative L2 error, and every element against the tensor's largest |gradient| (the noise of a row comes from sums of products
    # |P| |dP| that cancel in dS = P (dP - delta): it scales with the tensor's magnitude, not with the element's or the row's own)
    for name, got, ref in (("dq", N(dq), (to_tok(q.grad) * 0.125).numpy()), ("dk", N(dkv[:, :h * 64]), to_tok(k.grad).numpy()),
                           ("dv", N(dkv[:, h * 64:]), to_tok(v.grad).numpy())):
        rel, worst = attn_bwd_errors(got, ref)
        assert rel <= ATTN_BWD_REL_L2, (name, rel, worst)
        assert worst <= ATTN_BWD_ELEM, (name, rel, worst)
        # keys past k_len: exactly zero gradient (dq has analytically-zero rows too - a causal query that sees one key - which the
        # bf16 kernel only reproduces to rounding)
        if name != "dq":
            assert np.all(got[(np.abs(ref).max(axis=1) == 0)] == 0), name


ATTN_BWD_REL_L2 = 8e-3       # per tensor: ||got - ref|| / ||ref||   (measured 2.2-3.9e-3 over the nine shapes: tools/attn_bwd_stats.py)
ATTN_BWD_ELEM = 1.5e-2       # per element: |got - ref| / max |ref| over the tensor   (measured 2.1-6.8e-3)


def attn_bwd_errors(got, ref):
    rel = float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-30))
    worst = float(np.abs(got - ref).max() / max(np.abs(ref).max(), 1e-30))
    return rel, worst


def test_adam_step_matches_torch():
    g = torch.Generator().manual_seed(0)
    p0 = torch.randn(10000, generator=g)
    tp = torch.nn.Parameter(p0.clone())
    opt = torch.optim.Adam([tp], lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    p = p0.clone().to(DEV)
    m, v = torch.zeros_like(p), torch.zeros_like(p)
    p16 = torch.zeros(10000, device=DEV, dtype=torch.bfloat16)
    for step in range(1, 4):
        gr = torch.randn(10000, generator=g)
        tp.grad = gr.clone()
        opt.step()
        ops.adam_step(p, gr.to(DEV), m, v, 1e-3, 0.9, 0.98, 1e-9, step, p16=p16)
    np.testing.assert_allclose(N(p), tp.detach().numpy(), atol=1e-6, rtol=1e-5)
    np.testing.assert_array_equal(N(p16), N(p.bfloat16()))


@pytest.mark.parametrize("pfx", ["h", "r"])
def test_cif_backward_matches_reference_autograd(golden_dir, pfx):
    """d_hidden / d_alpha of cif() against the reference's autograd through its Python loop (tests/golden/g3_cif.npz)."""
    import os
    z = np.load(os.path.join(golden_dir, "g3_cif.npz"))
    alpha = torch.from_numpy(z[f"{pfx}_alpha"]).to(DEV)
    hidden = torch.from_numpy(z[f"{pfx}_hidden"]).to(DEV)
    w = torch.from_numpy(z[f"{pfx}_w"]).to(DEV)          # loss = (out * w).sum()  ->  d_out = w
    cur, rem, fire_idx, n_fire, n_label, tok = ops.cif_scan(alpha, 0.95, want_tok=True)
    d_hidden, d_alpha = ops.cif_bwd(hidden, cur, rem, tok, n_fire, w)
    np.testing.assert_allclose(N(d_hidden), z[f"{pfx}_ghidden"], atol=1e-5, rtol=1e-5)
    np.testing.assert_allclose(N(d_alpha), z[f"{pfx}_galpha"], atol=2e-5, rtol=1e-4)


@pytest.mark.parametrize("M,N_,K", [(1000, 2048, 256), (130, 256, 64), (4096, 512, 512)])
def test_relu_sign_bits_round_trip(M, N_, K):
    """FFN backward without re-reading the hidden activation: the first GEMM's epilogue writes one sign bit per ReLU output and the
    hidden gradient's GEMM masks from those bits - bit-identical to masking from the bf16 activation itself."""
    g = torch.Generator().manual_seed(M + K)
    x = torch.randn(M, K, generator=g).to(DEV).bfloat16()
    w1 = (torch.randn(N_, K, generator=g) / K ** 0.5).to(DEV).bfloat16()
    b1 = torch.randn(N_, generator=g).to(DEV)
    bits = ops.relu_bits_buffer(M, N_, DEV).zero_()
    hid = ops.gemm_nt_ex(x, w1, b1, out_dtype=torch.bfloat16, relu=True, relu_bits_out=bits)
    hid_plain = ops.gemm_nt(x, w1, b1, out_dtype=torch.bfloat16, relu=True)
    assert torch.equal(hid, hid_plain)
    # decode the image (asr_hip.h): [m/64][n/64][m & 7][n/8 & 7][m/8 & 7] bytes, bit n & 7
    Mp = bits.shape[0]
    img = bits.view(Mp // 64, N_ // 64, 8, 8, 8)                              # [bm, bn, m&7, n/8&7, m/8&7]
    rows = img.permute(0, 4, 2, 1, 3).reshape(Mp, N_ // 8)                    # [bm, m/8&7, m&7 | bn, n/8&7] = row-major [m, n/8]
    want = (hid.float() > 0).view(M, N_ // 8, 8).to(torch.uint8)
    packed = (want << torch.arange(8, device=DEV, dtype=torch.uint8)).sum(-1).to(torch.uint8)
    assert torch.equal(rows[:M], packed) and int(rows[M:].sum()) == 0
    dy = torch.randn(M, 256, generator=g).to(DEV).bfloat16()
    w2 = (torch.randn(256, N_, generator=g) / 16).to(DEV).bfloat16()
    a = ops.gemm_nn(dy, w2, out_dtype=torch.bfloat16, relu_mask=hid)
    b = ops.gemm_nn(dy, w2, out_dtype=torch.bfloat16, relu_bits=bits)
    assert torch.equal(a, b)


@pytest.mark.parametrize("shape", [(2, 21, 42, 10, 40), (3, 9, 42, 4, 40), (1, 501, 42, 250, 40), (2, 13, 20, 6, 18)])
def test_direct_conv_layer_backward_kernels(shape):
    """asr_conv_sub1_bwd_x / asr_conv_sub1_bwd_w (32 -> 32 channels, 3x3, stride (2, 1)) and asr_conv_sub0_bwd_w (1 -> 32) against torch
    autograd of F.conv2d on the same bf16-rounded operands: chunk tails (Tout not a multiple of 4 / 8), the last input row
    (Tin = 2 Tout + 1), narrow feature maps and the S2 shape's row length."""
    import torch.nn.functional as F
    from asr_amd import ops
    B, Tin, Fin, Tout, Fout = shape
    g = torch.Generator().manual_seed(Tin)
    x = torch.randn(B, Tin, Fin, 32, generator=g).bfloat16()            # the layer's input activation (ReLU output of the previous layer)
    x = torch.where(torch.rand(B, Tin, Fin, 32, generator=g) < 0.3, torch.zeros_like(x), x.abs())
    w = (torch.randn(32, 32, 3, 3, generator=g) * 0.1)
    dy = torch.randn(B, Tout, Fout, 32, generator=g).bfloat16()
    # reference: y = conv(x) on the [Tout, Fout] window the kernels see; gradient wrt x masked by relu'(x) like the layer below
    xr = x.float().permute(0, 3, 1, 2).requires_grad_(True)            # [B, C, T, F]
    wr = w.bfloat16().float().requires_grad_(True)
    y = F.conv2d(xr, wr, stride=(2, 1))[:, :, :Tout, :Fout]
    y.backward(dy.float().permute(0, 3, 1, 2))
    dx_ref = (xr.grad.permute(0, 2, 3, 1) * (x.float() > 0)).numpy()
    dw_ref = wr.grad.numpy()                                            # [co, ci, kh, kw]
    db_ref = dy.float().sum((0, 1, 2)).numpy()
    dxg = ops.conv_sub1_bwd_x(dy.to(DEV), w.to(DEV), x.to(DEV), Tout, Fout)
    np.testing.assert_allclose(dxg.float().cpu().numpy(), dx_ref, atol=3e-2 * np.abs(dx_ref).max(), rtol=2e-2)
    db = torch.zeros(32, device=DEV)
    dwm = ops.conv_sub1_bwd_w(dy.to(DEV), x.to(DEV), Tout, Fout, db=db).cpu().numpy()          # [co, tap*32 + ci]
    got = dwm.reshape(32, 9, 32).transpose(0, 2, 1).reshape(32, 32, 3, 3)
    np.testing.assert_allclose(got, dw_ref, atol=2e-3 * np.abs(dw_ref).max(), rtol=2e-3)
    np.testing.assert_allclose(db.cpu().numpy(), db_ref, atol=1e-3 * np.abs(db_ref).max() + 1e-3, rtol=1e-3)
    # the first layer: features [B, T, D] f32 (one channel), dy0 on [T1, F1]
    T, D, T1, F1 = Tin, Fin + 3, Tout, Fout
    feats = torch.randn(B, T, D, generator=g)
    fr = feats.bfloat16().float()[:, None].requires_grad_(False)
    w0 = torch.zeros(32, 1, 3, 3, requires_grad=True)
    y0 = F.conv2d(F.pad(fr, (0, 2, 0, 2)), w0, stride=(2, 1))[:, :, :T1, :F1]
    y0.backward(dy.float().permute(0, 3, 1, 2))
    dw0 = torch.zeros(32, 1, 3, 3, device=DEV)
    db0 = torch.zeros(32, device=DEV)
    ops.conv_sub0_bwd_w(dy.to(DEV), feats.to(DEV), dw0, db0, T1, F1)
    np.testing.assert_allclose(dw0.cpu().numpy(), w0.grad.numpy(), atol=3e-3 * np.abs(w0.grad.numpy()).max(), rtol=3e-3)
    np.testing.assert_allclose(db0.cpu().numpy(), db_ref, atol=1e-3 * np.abs(db_ref).max() + 1e-3, rtol=1e-3)
