"""The training step (forward -> ctc+ce -> HIP backward tape -> fused Adam/Noam) against the REFERENCE's own gradients and
optimizer step for the same model/batch (tests/golden/g1_ctc_transformer.npz, generated from /root/reference)."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from weights import make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_ctc_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return z, sd, model.to(DEV).eval()


def test_gradients_match_reference(golden_dir):
    z, sd, model = build(golden_dir)
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=5e-3)
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=5e-3)
    worst = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.detach().float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        worst.append((err / max(rn, 1e-12), err, name))
    # bf16 MFMA operands end to end vs the fp32 reference: each parameter's gradient is within 6 % relative L2, or - for
    # gradients that are (nearly) zero analytically, e.g. every w_ks.bias (softmax is shift-invariant per query) - within
    # 5e-3 absolute L2, two to three orders below the neighbouring gradients' norms.  Median relative error < 2 %.
    bad = [(r, e, n) for r, e, n in worst if r >= 6e-2 and e >= 5e-3]
    assert not bad, bad
    assert np.median([w[0] for w in worst]) < 2.5e-2


def test_optimizer_step_matches_reference(golden_dir):
    z, sd, model = build(golden_dir)
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    before = {k: p.detach().clone() for k, p in model.named_parameters()}
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    tr.step(x, lens, tg)
    torch.cuda.synchronize()
    np.testing.assert_allclose(tr.lr(), z["lr_step1"], rtol=1e-12)
    lr = float(z["lr_step1"])
    for key in [k for k in z.files if k.startswith("delta:")]:
        name = key[6:]
        p = dict(model.named_parameters())[name]
        got = (p.detach() - before[name]).float().cpu().numpy()
        ref = z[key]
        # step 1 of Adam moves every element by -lr*g/(|g|+eps): exactly reproduced from OUR gradient ...
        g = p.grad.float().cpu().numpy()
        # (p_after - p_before is only known to half an ulp of |p|, which for |p| ~ 1 is a visible fraction of lr ~ 8e-7)
        exp = -lr * g / (np.abs(g) + 1e-9)
        assert np.all(np.abs(got - exp) <= 1.2e-7 * np.abs(before[name].float().cpu().numpy()) + 2e-3 * lr + 1e-9), name
        # ... and equal to the reference's step wherever the fp32 gradient is not within bf16 noise of zero
        rg = z["grad:" + name].astype(np.float32)
        solid = np.abs(rg) > 0.2 * np.abs(rg).mean()
        assert np.mean(np.abs(got - ref)[solid] > 0.3 * lr) < 0.03, name
    # the bf16 shadow the MFMA kernels read is the rounded fp32 master
    np.testing.assert_array_equal(tr.fp.flat16.float().cpu().numpy(), tr.fp.flat.bfloat16().float().cpu().numpy())
    # a second step runs on the refreshed shadow and keeps the loss finite and decreasing-ish
    c2, e2 = tr.step(x, lens, tg)
    assert np.isfinite(float(c2)) and np.isfinite(float(e2))


def test_loss_decreases_over_steps(golden_dir):
    z, sd, model = build(golden_dir)
    # (k = 1.0 made this a chaotic regime: the weight-gradient kernels accumulate with float atomics, so runs differ in the last
    # bits, and 1 run in ~30 ended at 0.3-0.7 of the first loss instead of ~0.2; k = 0.5 trains as fast without the lottery)
    tr = asr_amd.Trainer(model, k=0.5, warmup_steps=20, label_smoothing=0.1)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    first = None
    for i in range(30):
        ctc, ce = tr.step(x, lens, tg, max_target_len=int((tg != 0).sum(1).max()))
        if i == 0:
            first = float(ctc) + float(ce)
    last = float(ctc) + float(ce)
    assert last < 0.8 * first, (first, last)


def test_conv_ctc_transformer_gradients_match_reference(golden_dir):
    """S2-family model (Conv2dSubsample front-end): conv / affine / encoder / decoder gradients vs the reference (G0)."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.Conv_CTC_Transformer.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=5e-3)
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=5e-3)
    params = dict(model.named_parameters())
    for key in [k for k in z.files if k.startswith("grad:")]:
        name = key[5:]
        ref, got = z[key], params[name].grad.float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        assert err < 6e-2 * rn or err < 5e-3, (name, err, rn)
    # every parameter's gradient norm (the fixture stores all of them, sorted by name)
    names = str(z["grad_names"]).split("|")
    for name, rn in zip(names, z["grad_norms"]):
        gn = float(params[name].grad.norm())
        assert abs(gn - rn) < 8e-2 * rn + 5e-3, (name, gn, rn)
    # and a few optimizer steps run end to end
    for _ in range(3):
        c2, e2 = tr.step(x, lens, tg)
    assert np.isfinite(float(c2)) and np.isfinite(float(e2))


def test_autograd_drop_in_matches_trainer_and_reference(golden_dir):
    """The reference's own loop shape (solver.py:83-93): model(...) -> cal_ctc_ce_loss -> loss.backward() -> torch optimizer."""
    z, sd, model = build(golden_dir)
    asr_amd.set_precision("bf16")
    model.train()
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    opt = torch.optim.Adam(model.parameters(), lr=1e-3, betas=(0.9, 0.98), eps=1e-9)
    l, ctc_logits, (logits, teos) = model(x, lens, tg)
    assert ctc_logits.requires_grad and logits.requires_grad
    ctc, ce = asr_amd.cal_ctc_ce_loss(ctc_logits, l, logits, teos, smoothing=0.1)
    loss = ctc + ce
    opt.zero_grad()
    loss.backward()
    worst = 0.0
    for name, p in model.named_parameters():
        assert p.grad is not None, name
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        assert err < 6e-2 * rn or err < 5e-3, (name, err, rn)
        worst = max(worst, err / max(rn, 1e-3))
    first = float(loss)
    for _ in range(20):
        opt.step()
        l, ctc_logits, (logits, teos) = model(x, lens, tg)
        ctc, ce = asr_amd.cal_ctc_ce_loss(ctc_logits, l, logits, teos, smoothing=0.1)
        loss = ctc + ce
        opt.zero_grad()
        loss.backward()
    assert float(loss) < 0.8 * first, (first, float(loss))
    # no_grad / eval inference still takes the plain path
    with torch.no_grad():
        l, c2, (lg2, _) = model(x, lens, tg)
    assert not c2.requires_grad


def test_cif_model_gradients_match_reference(golden_dir):
    """CIF family training step (solver.py:146-157): loss = 0.001 * qua + ctc + ce; every parameter's gradient vs the reference (G4),
    through the assigner (conv1d-as-GEMM + sigmoid), the alpha rescale, the integrate-and-fire accumulator and Decoder_CIF."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1, lambda_qua=0.001)
    x, lens, tg, noise = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets", "noise"))
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg, noise=noise)
    tr.backward(state)
    torch.cuda.synchronize()
    np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=5e-3)
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=1e-2)
    np.testing.assert_allclose(float(tr.last_qua), z["qua_loss"], rtol=2e-2)
    bad, rels = [], []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        rels.append(err / max(rn, 1e-12))
        # CIF chain (decoder -> integrate-and-fire -> assigner) on the tiny S0 model in bf16: per-tensor errors of 5-10 % on the
        # small-norm decoder attention weights move with the rounding realisation; tests/test_gpu_dropout.py::_grad_check
        # holds the whole gradient vector to 5 % as well
        if err >= 1.2e-1 * rn and err >= 5e-3:
            bad.append((name, err, rn))
    assert not bad, bad
    assert np.median(rels) < 4e-2
    # a few optimizer steps run end to end (noise drawn on device)
    for _ in range(3):
        c2, e2 = tr.step(x, lens, tg)
    assert np.isfinite(float(c2)) and np.isfinite(float(e2))


def test_cif_model_autograd_drop_in(golden_dir):
    """CIF solver loop shape (solver.py:146-157) through torch.autograd: loss = 0.001*qua + ctc + ce; .backward(); optimizer."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()   # G4 is an eval-mode fixture (the assigner's dropout defaults to 0.1 regardless of args.dropout)
    asr_amd.set_precision("bf16")
    x, lens, tg, noise = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets", "noise"))
    ctc_logits, l, num_pred, num, logits = model(x, lens, tg, noise=noise)
    qua, ctc, ce = asr_amd.cal_ctc_qua_ce_loss(ctc_logits, l, num_pred, num, logits, tg, smoothing=0.1)
    (0.001 * qua + ctc + ce).backward()
    bad = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        # CIF chain (decoder -> integrate-and-fire -> assigner) on the tiny S0 model in bf16: per-tensor errors of 5-10 % on the
        # small-norm decoder attention weights move with the rounding realisation; tests/test_gpu_dropout.py::_grad_check
        # holds the whole gradient vector to 5 % as well
        if err >= 1.2e-1 * rn and err >= 5e-3:
            bad.append((name, err, rn))
    assert not bad, bad


def test_ctc_model_gradients_match_reference(golden_dir):
    """ctcModel family (ctcModel/solver.py:30-36): loss = cal_loss(logits, len, targets); both the trainer's step and plain
    loss.backward() through the autograd bridge reproduce the reference's gradients (G5)."""
    from asr_amd.ctc_model import CTC_Model, Decoder, Encoder
    z = np.load(os.path.join(golden_dir, "g5_ctc_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    asr_amd.set_precision("bf16")

    def fresh():
        model = CTC_Model(Encoder(80, 2, 2, 64, 64, 64, 128, dropout=0.0, pe_maxlen=5000), Decoder(50, 64))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return model.to(DEV).train()

    def check(model):
        bad = []
        for name, p in model.named_parameters():
            ref = z["grad:" + name].astype(np.float32)
            got = p.grad.float().cpu().numpy()
            err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
            if err >= 6e-2 * rn and err >= 5e-3:
                bad.append((name, float(err), float(rn)))
        assert not bad, bad

    model = fresh()
    logits, l = model(x, lens)
    loss = asr_amd.cal_loss(logits, l, tg)
    np.testing.assert_allclose(float(loss), z["loss"], rtol=5e-3)
    loss.backward()
    check(model)
    model = fresh()
    tr = asr_amd.Trainer(model)
    tr.fp.grad.zero_()
    ctc, _, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    np.testing.assert_allclose(float(ctc), z["loss"], rtol=5e-3)
    check(model)
    tr.optimizer_step()


def test_ctc_side_stream_matches_serial_step(golden_dir):
    """The trainer queues the CTC branch on a side stream beside the decoder (Trainer._ctc_side_branch): same losses, and the same
    gradients up to fp32 summation order (the encoder-output gradient receives its CTC and decoder parts in a different order,
    and the weight-gradient kernels accumulate with float atomics) as the serial order."""
    z, sd, _ = build(golden_dir)
    asr_amd.set_precision("bf16")
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    grads, losses = [], []
    for overlap in (False, True):
        model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
        model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        model = model.to(DEV).eval()
        tr = asr_amd.Trainer(model, overlap_ctc=overlap)
        for _ in range(2):     # twice: the side stream's buffers get recycled
            tr.fp.grad.zero_()
            ctc, ce, st = tr.forward_loss(x, lens, tg)
            tr.backward(st)
        torch.cuda.synchronize()
        grads.append(tr.fp.grad.clone())
        losses.append((float(ctc), float(ce)))
    assert losses[0] == losses[1]
    np.testing.assert_allclose(grads[0].cpu().numpy(), grads[1].cpu().numpy(), rtol=2e-4, atol=2e-5)
