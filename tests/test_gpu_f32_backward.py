"""The backward tape in the fp32 parity mode (exact-f32 MFMA GEMMs, VALU attention forward / backward, f32 loss gradients):
every parameter's gradient against the REFERENCE's own autograd (fixtures G1, G5) at fp32 tolerance.  The bf16 product path is
compared with the same fixtures in test_gpu_trainer.py, but there bf16 operand rounding puts a 2-6 % floor under the comparison;
here a misplaced gradient, a wrong scale or a swapped accumulation in the tape's closures shows at 1e-4."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from weights import make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def grad_errors(model, z):
    out = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float64)
        got = p.grad.detach().double().cpu().numpy()
        out.append((float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref)), name))
    return out


def assert_f32_close(errs, rel=2e-4):
    # relative L2 <= 2e-4 per parameter, or - gradients that are analytically ~0 (w_ks.bias: softmax is shift-invariant) - 2e-6 absolute
    bad = [(e, r, n) for e, r, n in errs if e > rel * r and e > 2e-6]
    assert not bad, bad


def test_ctc_transformer_gradients_match_the_reference_in_f32(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_ctc_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    with asr_amd.precision("f32"):
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens, tg)
        tr.backward(state)
        torch.cuda.synchronize()
        np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=2e-5)
        np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=2e-5)
        errs = grad_errors(model, z)
        assert_f32_close(errs)
        # and the optimizer step that follows (solver.py:92-93, optimizer.py:19-29): the reference's parameter deltas
        before = {k: p.detach().clone() for k, p in model.named_parameters()}
        tr.optimizer_step()
        torch.cuda.synchronize()
        for key in [k for k in z.files if k.startswith("delta:")]:
            name = key[6:]
            got = (dict(model.named_parameters())[name].detach() - before[name]).double().cpu().numpy()
            ref = z[key].astype(np.float64)
            assert np.linalg.norm(got - ref) <= 2e-3 * np.linalg.norm(ref) + 1e-9, name


def test_ctc_model_gradients_match_the_reference_in_f32(golden_dir):
    from asr_amd.ctc_model import CTC_Model, Decoder, Encoder
    z = np.load(os.path.join(golden_dir, "g5_ctc_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    model = CTC_Model(Encoder(80, 2, 2, 64, 64, 64, 128, dropout=0.0, pe_maxlen=5000), Decoder(50, 64))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).train()
    with asr_amd.precision("f32"):
        tr = asr_amd.Trainer(model)
        tr.fp.grad.zero_()
        ctc, _, state = tr.forward_loss(x, lens, tg)
        tr.backward(state)
        np.testing.assert_allclose(float(ctc), z["loss"], rtol=2e-5)
        assert_f32_close(grad_errors(model, z))
    # the bf16 product path on the same fixture, for scale: its worst parameter is orders of magnitude further out
    model2 = CTC_Model(Encoder(80, 2, 2, 64, 64, 64, 128, dropout=0.0, pe_maxlen=5000), Decoder(50, 64))
    model2.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model2 = model2.to(DEV).train()
    with asr_amd.precision("bf16"):
        tr2 = asr_amd.Trainer(model2)
        tr2.fp.grad.zero_()
        _, _, state = tr2.forward_loss(x, lens, tg)
        tr2.backward(state)
        worst16 = max(e / max(r, 1e-12) for e, r, n in grad_errors(model2, z) if r > 1e-3)
    assert worst16 > 1e-3          # (the floor the f32 mode removes)


def test_conv_ctc_transformer_gradients_match_the_reference_in_f32(golden_dir):
    """G0: the conv front end's backward too (patch-matrix GEMMs, col2im, the permuted affine weight gradient)."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.Conv_CTC_Transformer.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    with asr_amd.precision("f32"):
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens, tg)
        tr.backward(state)
        torch.cuda.synchronize()
    np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=2e-5)
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=2e-5)
    params = dict(model.named_parameters())
    errs = []
    for key in [k for k in z.files if k.startswith("grad:")]:
        ref, got = z[key].astype(np.float64), params[key[5:]].grad.double().cpu().numpy()
        errs.append((float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref)), key[5:]))
    assert_f32_close(errs)
    names = str(z["grad_names"]).split("|")          # every parameter's gradient norm
    for name, rn in zip(names, z["grad_norms"]):
        gn = float(params[name].grad.norm())
        assert abs(gn - rn) < 3e-4 * rn + 2e-6, (name, gn, rn)


def test_cif_model_gradients_match_the_reference_in_f32(golden_dir):
    """G4: loss = 0.001 * qua + ctc + ce through the assigner, the alpha rescale, the integrate-and-fire accumulator and Decoder_CIF."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    model = model.to(DEV).eval()
    x, lens, tg, noise = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets", "noise"))
    with asr_amd.precision("f32"):
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1, lambda_qua=0.001)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens, tg, noise=noise)
        tr.backward(state)
        torch.cuda.synchronize()
    np.testing.assert_allclose(float(ctc), z["ctc_loss"], rtol=5e-5)
    np.testing.assert_allclose(float(ce), z["ce_loss_s01"], rtol=5e-5)
    np.testing.assert_allclose(float(tr.last_qua), z["qua_loss"], rtol=5e-5)
    # (one tensor, conv_encoder.affine.weight, sits at 2.06e-4: fp32 summation order over the B*L rows that feed three consumers here)
    assert_f32_close(grad_errors(model, z), rel=4e-4)
