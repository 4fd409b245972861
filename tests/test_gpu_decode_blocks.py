"""The fused sub-layers of the per-token decode step (decode_blocks.hip) against stock torch in f32 on the same bf16-rounded operands:
PositionwiseFeedForward (src/transformer/module.py:48-53) and MultiheadAttention for one new position against a K / V cache
(src/transformer/attention.py:33-62)."""
import numpy as np
import pytest
import torch
import torch.nn.functional as F

from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def rnd(g, *shape, scale=1.0):
    return (torch.randn(*shape, generator=g) * scale)


@pytest.mark.parametrize("M", [1, 16, 33, 160])
def test_decode_ffn(M):
    g = torch.Generator().manual_seed(M)
    x = rnd(g, M, 256)
    w1, b1 = rnd(g, 2048, 256, scale=256 ** -0.5), rnd(g, 2048, scale=0.1)
    w2, b2 = rnd(g, 256, 2048, scale=2048 ** -0.5), rnd(g, 256, scale=0.1)
    gamma, beta = 1.0 + rnd(g, 256, scale=0.1), rnd(g, 256, scale=0.1)
    xb, w1b, w2b = x.bfloat16(), w1.bfloat16(), w2.bfloat16()
    hid = torch.relu(xb.float() @ w1b.float().t() + b1).bfloat16().float()                       # the kernel keeps the hidden row in bf16
    ref = F.layer_norm(hid @ w2b.float().t() + b2 + x, (256,), gamma, beta, 1e-5)
    args = [t.to(DEV).contiguous() for t in (xb, x, w1b, b1, w2b, b2, gamma, beta)]
    for _ in range(3):                                                                            # the workspace comes back zeroed
        y32, y16 = ops.decode_ffn(*args, 1e-5)
        np.testing.assert_allclose(y32.cpu().numpy(), ref.numpy(), rtol=0, atol=6e-3)
        np.testing.assert_allclose(y16.float().cpu().numpy(), ref.numpy(), rtol=1e-2, atol=2e-2)
    assert int(ops.decode_block_workspace(M, torch.device(DEV)).view(torch.int32).abs().sum()) == 0       # comes back zeroed


@pytest.mark.parametrize("M,h,Tmax,t", [(1, 4, 64, 0), (16, 4, 64, 5), (33, 4, 64, 63), (40, 4, 128, 64), (160, 4, 100, 99), (7, 2, 32, 17)])
def test_decode_self_attn(M, h, Tmax, t):
    g = torch.Generator().manual_seed(1000 * M + t)
    D = 256
    x = rnd(g, M, D)
    wq, wk, wv = (rnd(g, h * 64, D, scale=D ** -0.5) for _ in range(3))
    bq, bk, bv = (rnd(g, h * 64, scale=0.1) for _ in range(3))
    wo, bo = rnd(g, D, h * 64, scale=(h * 64) ** -0.5), rnd(g, D, scale=0.1)
    gamma, beta = 1.0 + rnd(g, D, scale=0.1), rnd(g, D, scale=0.1)
    kc, vc = rnd(g, M, h, Tmax, 64).bfloat16(), rnd(g, M, h, Tmax, 64).bfloat16()
    xb = x.bfloat16()
    wqkv = torch.cat([wq, wk, wv], 0).bfloat16()
    bqkv = torch.cat([bq, bk, bv], 0)
    wob = wo.bfloat16()
    # reference
    proj = xb.float() @ wqkv.float().t() + bqkv
    q, k, v = (proj[:, i * h * 64:(i + 1) * h * 64].view(M, h, 64) for i in range(3))
    k_new, v_new = k.bfloat16(), v.bfloat16()
    K = torch.cat([kc[:, :, :t].float(), k_new.float()[:, :, None]], 2)                            # [M, h, t + 1, 64]
    V = torch.cat([vc[:, :, :t].float(), v_new.float()[:, :, None]], 2)
    att = torch.softmax((q[:, :, None] @ K.transpose(2, 3)) / 8.0, -1)                             # [M, h, 1, t + 1]
    o = (att @ V).reshape(M, h * 64).bfloat16().float()                                            # the kernel feeds the output projection in bf16
    ref = F.layer_norm(o @ wob.float().t() + bo + x, (D,), gamma, beta, 1e-5)
    state = torch.tensor([t, -1], dtype=torch.int32, device=DEV)
    kcd, vcd = kc.to(DEV).contiguous(), vc.to(DEV).contiguous()
    args = [a.to(DEV).contiguous() for a in (xb, x, wqkv, bqkv, wob, bo, gamma, beta)]
    y32, y16 = ops.decode_self_attn(*args, kcd, vcd, state, 1e-5)
    np.testing.assert_allclose(y32.cpu().numpy(), ref.numpy(), rtol=0, atol=1.5e-2)
    np.testing.assert_allclose(kcd[:, :, t].float().cpu().numpy(), k_new.float().numpy(), rtol=8e-3, atol=1e-5)      # one bf16 ulp
    np.testing.assert_allclose(vcd[:, :, t].float().cpu().numpy(), v_new.float().numpy(), rtol=8e-3, atol=1e-5)
    other = [p for p in range(Tmax) if p != t]
    np.testing.assert_array_equal(kcd[:, :, other].float().cpu().numpy(), kc[:, :, other].float().numpy())      # nothing else is touched
    y32b, _ = ops.decode_self_attn(*args, kcd, vcd, state, 1e-5)                                   # same slot again: same result
    np.testing.assert_allclose(y32b.cpu().numpy(), y32.cpu().numpy(), rtol=0, atol=2e-6)          # (the heads' partial rows meet in float atomics)
    assert int(ops.decode_block_workspace(M, torch.device(DEV)).view(torch.int32).abs().sum()) == 0


@pytest.mark.parametrize("M", [1, 32, 50])
def test_decode_ffn_with_the_attention_output_prologue(M):
    """x = LayerNorm0(ctx Wo^T + bo + res) (attention.py:58-60) computed inside the feed-forward launch"""
    g = torch.Generator().manual_seed(77 + M)
    ctx, res = rnd(g, M, 256), rnd(g, M, 256)
    wo, bo = rnd(g, 256, 256, scale=256 ** -0.5), rnd(g, 256, scale=0.1)
    g0, bt0 = 1.0 + rnd(g, 256, scale=0.1), rnd(g, 256, scale=0.1)
    w1, b1 = rnd(g, 2048, 256, scale=256 ** -0.5), rnd(g, 2048, scale=0.1)
    w2, b2 = rnd(g, 256, 2048, scale=2048 ** -0.5), rnd(g, 256, scale=0.1)
    gamma, beta = 1.0 + rnd(g, 256, scale=0.1), rnd(g, 256, scale=0.1)
    cb, wob, w1b, w2b = ctx.bfloat16(), wo.bfloat16(), w1.bfloat16(), w2.bfloat16()
    x = F.layer_norm(cb.float() @ wob.float().t() + bo + res, (256,), g0, bt0, 1e-5)
    hid = torch.relu(x.bfloat16().float() @ w1b.float().t() + b1).bfloat16().float()
    ref = F.layer_norm(hid @ w2b.float().t() + b2 + x, (256,), gamma, beta, 1e-5)
    d = lambda t: t.to(DEV).contiguous()
    for _ in range(2):
        y32, y16 = ops.decode_ffn(d(cb), d(res), d(w1b), d(b1), d(w2b), d(b2), d(gamma), d(beta), 1e-5, pre=(d(wob), d(bo), d(g0), d(bt0), 1e-5))
        np.testing.assert_allclose(y32.cpu().numpy(), ref.numpy(), rtol=0, atol=1.2e-2)
    assert int(ops.decode_block_workspace(M, torch.device(DEV)).view(torch.int32).abs().sum()) == 0


def test_fused_sub_layers_equal_the_separate_launches_on_module_weights(monkeypatch):
    """module level (a d_model = 256 decoder's own parameters: q | k | v concatenation, output projection, biases, LayerNorms): the
    one-launch sub-layers against the separate launches they replace, for the self-attention step and for cross-attention +
    feed-forward, greedy (Lq = 1) and beam (Lq = beam) row labellings; then a whole greedy decode, fused against separate"""
    import asr_amd
    from asr_amd import modules
    from asr_amd.modules import Act
    torch.manual_seed(3)
    dec = asr_amd.Decoder(2, 3, 40, 2, 4, 256, 512, dropout=0.1).to(DEV).eval()
    g = torch.Generator().manual_seed(11)
    B, beam, L, Tmax, t = 3, 2, 70, 16, 6
    N = B * beam
    enc = Act(rnd(g, B * L, 256).to(DEV), None, B, L)
    enc_len = torch.tensor([70, 33, 51], dtype=torch.int32, device=DEV)
    state = torch.tensor([t, -1], dtype=torch.int32, device=DEV)
    k_len = torch.full((N,), t + 1, dtype=torch.int32, device=DEV)
    x32 = rnd(g, N, 256).to(DEV)
    out = {}
    with asr_amd.precision("bf16"), torch.no_grad():
        cross = dec._cross_kv(enc)
        for fused in (True, False):
            monkeypatch.setattr(modules, "_DECODE_FUSED", fused)
            with modules._decode_step():
                kc = (rnd(torch.Generator().manual_seed(5), N, 4, Tmax, 64)).bfloat16().to(DEV)
                vc = (rnd(torch.Generator().manual_seed(6), N, 4, Tmax, 64)).bfloat16().to(DEV)
                x = Act(x32.clone(), x32.bfloat16(), N, 1)
                layer = dec.layer_stack[0]
                y = layer.slf_attn._impl_cached_self(x, kc, vc, state, k_len, next_attn=layer.enc_attn, next_lq=beam)
                if fused:      # the cross attention's queries produced by the same launch == asr_proj_heads on the output rows
                    import math
                    att = layer.enc_attn
                    qref = ops.proj_heads(y.b16, att._w("q", (att.w_qs.weight,)), att._b("bq", (att.w_qs.bias,)), 1, B, beam, 4,
                                          modules._LOG2E / math.sqrt(64))[0]
                    assert tuple(y.next_q.shape) == tuple(qref.shape)
                    np.testing.assert_allclose(y.next_q.float().cpu().numpy(), qref.float().cpu().numpy(), rtol=2e-2, atol=2e-2)
                z1 = layer._decode_cross_ffn(Act(y.f32, y.b16, N, 1), Act(enc.f32.view(B, 1, L, 256).expand(B, beam, L, 256).reshape(N * L, 256).contiguous(), None, N, L),
                                             enc_len.repeat_interleave(beam), tuple(None if c is None else c.repeat_interleave(beam, 0).contiguous() for c in cross(0)[:2]) + (None,), N)
                z2 = layer._decode_cross_ffn(Act(y.f32, y.b16, B, beam), enc, enc_len, cross(0), N)
            out[fused] = (y.f32.float().cpu().numpy(), kc[:, :, t].float().cpu().numpy(), z1.f32.cpu().numpy(), z2.f32.cpu().numpy())
    for a, b in zip(out[True], out[False]):
        np.testing.assert_allclose(a, b, rtol=2e-2, atol=3e-2)
    np.testing.assert_allclose(out[True][2], out[True][3], rtol=0, atol=2e-2)          # beams as query positions == beams as batch rows
    # whole greedy decode on this model: the fused step and the separate launches agree on (nearly) every token - bf16 paths that
    # differ in rounding order may part ways at a near-tie, after which the prefixes differ, so only the first tokens are compared
    encoded, lens = rnd(g, B, L, 256).to(DEV), enc_len.long()
    res = {}
    with asr_amd.precision("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(modules, "_DECODE_FUSED", fused)
            dec.__dict__.pop("_decode_graph", None)
            res[fused] = dec.batch_decode(encoded, lens, max_decode_len=10)[0].cpu().numpy()
    assert res[True].shape == res[False].shape
    assert (res[True][:, :3] == res[False][:, :3]).mean() >= 0.75


def test_beam_searches_with_fused_sub_layers(monkeypatch):
    """d_model = 256 models (the fused launches' shape): Decoder.batch_beam_decode and Decoder_CIF.batch_recognize_beam with the
    one-launch sub-layers against the separate launches - same shapes and lengths, first tokens of the best beam equal, scores close
    (bf16 paths with different rounding order may part ways later at a near-tie)"""
    import asr_amd
    from asr_amd import modules
    torch.manual_seed(7)
    g = torch.Generator().manual_seed(21)
    B, L, beam, T = 3, 60, 3, 8
    dec = asr_amd.Decoder(2, 3, 40, 2, 4, 256, 512, dropout=0.1).to(DEV).eval()
    encoded, lens = rnd(g, B, L, 256).to(DEV), torch.tensor([60, 41, 17], device=DEV)
    dcif = asr_amd.Decoder_CIF(2, 40, 2, 4, 256, 512, dropout=0.1).to(DEV).eval()
    frames, n_frames = rnd(g, B, 9, 256).to(DEV), torch.tensor([9, 4, 6], dtype=torch.int32, device=DEV)
    res = {}
    with asr_amd.precision("bf16"):
        for fused in (True, False):
            monkeypatch.setattr(modules, "_DECODE_FUSED", fused)
            dec.__dict__.pop("_beam_graph", None)
            dcif.__dict__.pop("_beam_graph", None)
            p, l, sc = dec.batch_beam_decode(encoded, lens, beam_size=beam, max_decode_len=T)
            hyp = dcif.batch_recognize_beam(frames, n_frames, beam, nbest=2)
            res[fused] = (p.cpu().numpy(), l.cpu().numpy(), sc.cpu().numpy(), hyp)
    a, b = res[True], res[False]
    assert a[0].shape == b[0].shape and a[1].shape == b[1].shape
    assert (a[0][:, 0, :2] == b[0][:, 0, :2]).mean() >= 0.66
    np.testing.assert_allclose(a[2][:, 0], b[2][:, 0], rtol=0, atol=0.15)
    for (ya, la), (yb, lb), n in zip(a[3], b[3], (9, 4, 6)):
        assert la == lb == [n + 1, n + 1] and ya[0][:2] == yb[0][:2]
