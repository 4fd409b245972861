"""The attention-only family end to end: `Transformer` (src/transformer/transformer.py:21-35) stepped the way
`Transformer_Solver` does it (src/transformer/solver.py:26-35: loss = cal_ce_loss(logits, targets_eos) alone), against fixture
G17 (tests/golden/g17_transformer.npz: the reference's logits, loss, every gradient and its optimizer step)."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from weights import crc_of, make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(golden_dir, train=False):
    z = np.load(os.path.join(golden_dir, "g17_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    assert crc_of(sd) == int(z["crc"])
    model = asr_amd.Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
    missing = model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    assert not missing.unexpected_keys and all(k.endswith("positional_encoding.pe") for k in missing.missing_keys)
    # the state_dict contract of SURVEY §8(b): the plain family has no ctc_fc
    assert not any(k.startswith("ctc_fc") for k in model.state_dict())
    model = model.to(DEV)
    return z, sd, (model.train() if train else model.eval())


def batch(z):
    return tuple(torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))


@pytest.mark.parametrize("prec", ["f32", "bf16"])
def test_forward_returns_the_references_two_tuple(golden_dir, prec):
    z, sd, model = build(golden_dir)
    x, lens, tg = batch(z)
    with asr_amd.precision(prec), torch.no_grad():
        out = model(x, lens, tg)
        assert isinstance(out, tuple) and len(out) == 2          # (logits, targets_eos), transformer.py:35
        logits, teos = out
        enc = model.encoder(x, lens)
        ce = asr_amd.cal_ce_loss(logits, teos, smoothing=0.1)
        ce0 = asr_amd.cal_ce_loss(logits, teos, smoothing=0.0)
    tol = dict(atol=5e-4, rtol=1e-3) if prec == "f32" else dict(atol=3e-2, rtol=2e-2)
    np.testing.assert_array_equal(teos.cpu().numpy(), z["targets_eos"])
    np.testing.assert_allclose(enc.float().cpu().numpy(), z["enc_out"], **tol)
    np.testing.assert_allclose(logits.float().cpu().numpy(), z["logits"], **tol)
    ltol = 1e-4 if prec == "f32" else 5e-3
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=ltol)
    np.testing.assert_allclose(float(ce0), z["ce_loss_s0"], rtol=ltol)
    e = enc.float().cpu().numpy()
    for b, n in enumerate(z["lens"]):
        assert np.all(e[b, n:] == 0)            # encoder.py:74,77


def test_every_gradient_and_the_solver_step_in_f32(golden_dir):
    z, sd, model = build(golden_dir)
    x, lens, tg = batch(z)
    with asr_amd.precision("f32"):
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens, tg)
        tr.backward(state)
        torch.cuda.synchronize()
        assert float(ctc) == 0.0                 # no CTC term in this family (solver.py:28-31)
        np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=2e-5)
        bad = []
        for name, p in model.named_parameters():
            ref = z["grad:" + name].astype(np.float64)
            got = p.grad.detach().double().cpu().numpy()
            e, r = float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref))
            if e > 2e-4 * r and e > 2e-6:
                bad.append((e, r, name))
        assert not bad, bad
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        tr.optimizer_step()
        torch.cuda.synchronize()
        np.testing.assert_allclose(tr.lr(), z["lr_step1"], rtol=1e-12)
        for key in [k for k in z.files if k.startswith("delta:")]:
            name = key[6:]
            got = (dict(model.named_parameters())[name].detach() - before[name]).double().cpu().numpy()
            ref = z[key].astype(np.float64)
            assert np.linalg.norm(got - ref) <= 2e-3 * np.linalg.norm(ref) + 1e-9, name
        # the reference's loss after its step, from a second forward of the stepped model
        with torch.no_grad():
            logits2, teos2 = model(x, lens, tg)
            ce2 = asr_amd.cal_ce_loss(logits2, teos2, smoothing=0.1)
        np.testing.assert_allclose(float(ce2), z["ce_loss_s01_after_step"], rtol=2e-5)


def test_bf16_trainer_steps_the_two_tuple_model(golden_dir):
    z, sd, model = build(golden_dir)
    x, lens, tg = batch(z)
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=5e-3)
    worst = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.detach().float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        worst.append((err / max(rn, 1e-12), err, name))
    # bf16 operands end to end against the fp32 reference.  Without a CTC branch the encoder's whole gradient arrives through the
    # decoder's cross attention (two bf16 attention backwards deep), so this fixture's floor is higher than G1's (median 5.0 % against
    # 1.3 %, tools/g17_bf16_err.py); the tape itself is held to 2e-4 by the f32 test above.  Per parameter: relative L2 < 12 %, or - for
    # gradients that are (nearly) zero analytically, every w_ks.bias (softmax is shift-invariant per query) - absolute L2 < 5e-3.
    bad = [(r, e, n) for r, e, n in worst if r >= 1.2e-1 and e >= 5e-3]
    assert not bad, bad
    assert np.median([w[0] for w in worst]) < 7e-2
    # and full steps: loss decreases on the fixture's batch with a training-speed schedule
    tr2 = asr_amd.Trainer(build(golden_dir)[2], k=0.5, warmup_steps=20, label_smoothing=0.1)
    first = None
    for i in range(30):
        c, e = tr2.step(x, lens, tg, max_target_len=int((tg != 0).sum(1).max()))
        first = float(e) if i == 0 else first
    assert float(c) == 0.0 and float(e) < 0.8 * first, (first, float(e))


def test_autograd_bridge_matches_the_trainer(golden_dir):
    """The reference's own loop shape: loss = cal_ce_loss(...); loss.backward() (solver.py:31-34) through modules._TapeFn."""
    z, sd, model = build(golden_dir)
    x, lens, tg = batch(z)
    asr_amd.set_precision("bf16")
    model.zero_grad()
    logits, teos = model(x, lens, tg)
    loss = asr_amd.cal_ce_loss(logits, teos, smoothing=0.1)
    loss.backward()
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(loss), z["ce_loss_s01"], rtol=5e-3)
    for name in ("decoder.tgt_word_prj.weight", "encoder.layer_stack.0.slf_attn.w_qs.weight", "decoder.tgt_word_emb.weight"):
        ref = z["grad:" + name].astype(np.float32)
        got = dict(model.named_parameters())[name].grad.float().cpu().numpy()
        assert np.linalg.norm(got - ref) < 1.2e-1 * np.linalg.norm(ref), name      # (measured 0.6-7.6 %: tools/g17_bf16_err.py)



def test_the_two_tuple_step_replays_from_a_graph(golden_dir):
    """Trainer.step_graphed on the attention-only family (no CTC branch, so no side-branch fork / join in the captured step): the
    replayed steps stay on the eager trajectory."""
    z, sd, m_e = build(golden_dir, train=True)
    _, _, m_g = build(golden_dir, train=True)
    x, lens, tg = batch(z)
    umax = int((tg != 0).sum(1).max())
    asr_amd.set_precision("bf16")
    te = asr_amd.Trainer(m_e, k=0.2, warmup_steps=50, label_smoothing=0.1)
    tg_ = asr_amd.Trainer(m_g, k=0.2, warmup_steps=50, label_smoothing=0.1)
    le, lg = [], []
    for i in range(8):
        c, e = te.step(x, lens, tg, max_target_len=umax)
        le.append(float(e))
        c, e = tg_.step_graphed(x, lens, tg, max_target_len=umax)
        lg.append(float(e))
        assert float(c) == 0.0
    assert tg_.graph_active(), tg_._graph_failed
    np.testing.assert_allclose(np.array(lg), np.array(le), rtol=5e-3)
    pe, pg = te.fp.flat.float().cpu().numpy(), tg_.fp.flat.float().cpu().numpy()
    assert np.linalg.norm(pg - pe) / np.linalg.norm(pe) < 2e-3
