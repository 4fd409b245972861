"""The encoder self-attention forward's hand-scheduled form (csrc/attention_fwd4.hip + the generated attention_fwd4_asm.inc):
non-causal, Lq >= 128, through the C-ABI (asr_attention_fwd) against an fp64 softmax (attention.py:76-84).

The kernel keeps NO running maximum: scores are exponentiated against reference 0 until a lane's partial row sum leaves (2^-64, 2^64),
and only then a rare branch re-centres the rows.  cdna_hip_programming.md rule 26: a rare data-dependent branch needs inputs that
FORCE it - scores hundreds above and below zero, rows whose first tiles lie far below their maximum, a single spike key - and a full
independent reference.  Both workgroup forms (2 and 4 waves) and the dropout stream are covered."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
LOG2E = 1.4426950408889634

CASES = [(2, 2, 300, 300, False, 1.0, "plain"), (3, 4, 200, 200, True, 1.0, "plain"), (2, 4, 1000, 1000, True, 1.0, "plain"),
         (1, 8, 256, 64, False, 1.0, "plain"), (2, 2, 130, 1, False, 1.0, "plain"), (2, 2, 257, 5, True, 1.0, "plain"),
         (2, 2, 384, 129, False, 1.0, "plain"), (2, 2, 300, 300, False, 12.0, "wide"), (2, 4, 1000, 1000, True, 30.0, "wide"),
         (2, 2, 300, 300, False, 1.0, "ramp-up"), (2, 2, 300, 300, True, 1.0, "ramp-down"), (2, 2, 512, 512, False, 1.0, "low"),
         (2, 2, 512, 512, False, 1.0, "high"), (1, 2, 256, 640, False, 1.0, "spike"), (32, 4, 250, 250, True, 1.0, "plain")]


def _inputs(B, h, Lq, Lk, ragged, scale, kind):
    g = torch.Generator().manual_seed(B * 1000 + Lq + Lk)
    q = torch.randn(B, h, Lq, 64, generator=g) * 0.5 * scale
    k = torch.randn(B, h, Lk, 64, generator=g)
    v = torch.randn(B, h, Lk, 64, generator=g)
    if kind not in ("plain", "wide"):
        # dimension 0 carries an additive per-key offset: q[..., 0] = 1 (natural units), k[..., 0] = offset
        q[..., 0] = 1.0
        pos = torch.arange(Lk, dtype=torch.float32)
        k[..., 0] = {"ramp-up": pos * 1.5 - 300.0, "ramp-down": 250.0 - pos * 1.5, "low": torch.full((Lk,), -400.0),
                     "high": torch.full((Lk,), 300.0), "spike": torch.where(pos == 333, torch.tensor(500.0), torch.tensor(-100.0))}[kind]
    k_len = None
    if ragged:
        k_len = torch.randint(max(1, Lk // 2), Lk + 1, (B,), generator=g)
        k_len[0] = Lk
        if B > 1:
            k_len[1] = min(Lk, 3)
    return (q * LOG2E).to(DEV).bfloat16(), k.to(DEV).bfloat16(), v.to(DEV).bfloat16(), k_len


def _reference(qd, kd, vd, k_len, keep=None, dscale=1.0):
    B, h, Lq, _ = qd.shape
    Lk = kd.shape[2]
    s = qd.double().cpu() @ kd.double().cpu().transpose(-1, -2)          # base-2 logits
    if k_len is not None:
        s = s.masked_fill((torch.arange(Lk)[None, :] >= k_len[:, None])[:, None, None, :], float("-inf"))
    m = s.max(-1, keepdim=True).values
    p = torch.exp2(s - m)
    l = p.sum(-1, keepdim=True)
    p = p / l
    if keep is not None:
        p = p * keep.double() * dscale
    ctx = (p @ vd.double().cpu()).permute(0, 2, 1, 3).reshape(B, Lq, h * 64)
    return ctx, (m + torch.log2(l)).squeeze(-1), s


@pytest.mark.parametrize("B,h,Lq,Lk,ragged,scale,kind", CASES)
def test_forward_matches_fp64_softmax(B, h, Lq, Lk, ragged, scale, kind):
    qd, kd, vd, k_len = _inputs(B, h, Lq, Lk, ragged, scale, kind)
    ctx, lse = ops.attention_fwd(qd, kd, vd, None if k_len is None else k_len.to(DEV).int(), False, need_lse=True)
    ref, lse_ref, s = _reference(qd, kd, vd, k_len)
    if kind != "plain":
        assert float(s[torch.isfinite(s)].abs().max()) > 300.0           # the input really leaves the exp2 range
    assert bool(torch.isfinite(ctx).all())
    # bf16 probabilities and a bf16 output: 2^-9 relative on sums of |v| <= ~4
    np.testing.assert_allclose(ctx.double().cpu().numpy(), ref.numpy(), atol=2e-2, rtol=0)
    np.testing.assert_allclose(lse.double().cpu().numpy(), lse_ref.numpy(), rtol=2e-5, atol=2e-5)


@pytest.mark.parametrize("B,h,Lq,Lk,ragged,scale,kind", [(2, 4, 1000, 1000, True, 1.0, "plain"), (3, 4, 200, 200, True, 1.0, "plain"),
                                                         (2, 2, 300, 300, False, 12.0, "wide"), (1, 2, 256, 640, False, 1.0, "spike")])
def test_dropout_stream_uses_the_keep_bit_image(B, h, Lq, Lk, ragged, scale, kind):
    qd, kd, vd, k_len = _inputs(B, h, Lq, Lk, ragged, scale, kind)
    d = ops.Dropout(6554, 11, 22)
    bits = ops.attention_dropmask(d, B, h, Lq, Lk, DEV)
    ctx, lse = ops.attention_fwd(qd, kd, vd, None if k_len is None else k_len.to(DEV).int(), False, need_lse=True, drop=d, drop_bits=bits)
    # the Mk image (asr_common.h): words [bh][key / 32][query padded to 128], bit key & 31
    lqp, kw = (Lq + 127) // 128 * 128, ((Lk + 127) // 128 * 128) // 32
    mk = bits[:B * h * kw * lqp].view(B, h, kw, lqp).cpu().contiguous().numpy().view(np.uint32)
    keep = ((mk[:, :, :, :Lq, None] >> np.arange(32, dtype=np.uint32)) & 1).transpose(0, 1, 3, 2, 4).reshape(B, h, Lq, kw * 32)[..., :Lk]
    ref, lse_ref, _ = _reference(qd, kd, vd, k_len, keep=torch.from_numpy(keep.astype(np.float64)), dscale=65536.0 / (65536 - 6554))
    np.testing.assert_allclose(ctx.double().cpu().numpy(), ref.numpy(), atol=2.5e-2, rtol=0)
    np.testing.assert_allclose(lse.double().cpu().numpy(), lse_ref.numpy(), rtol=2e-5, atol=2e-5)       # the row sums are of ALL probabilities


def test_both_workgroup_forms_agree_bit_for_bit():
    """2 and 4 waves per workgroup run the same per-wave instruction stream on the same rows: identical outputs."""
    import subprocess, sys
    code = ("import torch, sys; sys.path.insert(0, %r); import asr_amd; from asr_amd import ops; torch.manual_seed(5);"
            "q = (torch.randn(2, 4, 600, 64) * 0.7).cuda().bfloat16(); k = torch.randn(2, 4, 700, 64).cuda().bfloat16(); v = torch.randn(2, 4, 700, 64).cuda().bfloat16();"
            "kl = torch.tensor([700, 333], dtype=torch.int32).cuda(); c, l = ops.attention_fwd(q, k, v, kl, False, need_lse=True);"
            "torch.save((c.cpu(), l.cpu()), sys.argv[1])") % os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for nw in ("2", "4"):
        path = "/tmp/attn4_nw%s.pt" % nw
        subprocess.run([sys.executable, "-c", code, path], check=True, env=dict(os.environ, ASR_AMD_ATTN_NW=nw))
        outs.append(torch.load(path))
    assert torch.equal(outs[0][0], outs[1][0]) and torch.equal(outs[0][1], outs[1][1])
