"""mask_lm on the HIP path against the reference's outputs (tests/golden/g11_mask_lm.npz: Mask_LM.py:19-63,
loss.py:5-45 and the gradients of both solver steps, solver.py:85-114 / :218-246)."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import mask_lm, ops
from weights import crc_of, make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(golden_dir):
    z = np.load(os.path.join(golden_dir, "g11_mask_lm.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    assert crc_of(sd) == int(z["crc"])
    model = mask_lm.Mask_LM(mask_lm.Encoder(int(z["n_src"]), 2, 2, int(z["d_model"]), 128, dropout=0.0),
                            mask_lm.Decoder(int(z["n_tgt"]), int(z["d_model"])))
    keys = [k for k, _ in names_shapes_from_json(z["names_shapes"])]
    assert list(model.state_dict().keys()) == keys                 # the reference's state_dict contract
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return z, model.to(DEV).train()


def rand_of_fixture(shape):
    torch.manual_seed(1111)                                        # the reference drew torch.rand((B, T)) on the CPU from this seed
    return torch.rand(shape)


def test_token_mask_and_pretraining_forward(golden_dir):
    z, model = build(golden_dir)
    ids, lens = torch.from_numpy(z["ids"]).to(DEV), torch.from_numpy(z["lens"]).to(DEV)
    rand = rand_of_fixture(z["ids"].shape)
    mi, mk = model.token_mask(ids, rand=rand.to(DEV))
    np.testing.assert_array_equal(mi.cpu().numpy(), z["masked_ids"])
    np.testing.assert_array_equal(mk.cpu().numpy(), z["masked_index"])
    for prec, tol in (("f32", dict(atol=5e-4, rtol=1e-3)), ("bf16", dict(atol=6e-2, rtol=2e-2))):
        with asr_amd.precision(prec), torch.no_grad():
            logits_AE, logits, mask = model(ids, lens, rand=rand.to(DEV))
            ce = mask_lm.cal_ce_mask_loss(logits_AE, ids, mask, smoothing=0.1)
        assert logits is None
        np.testing.assert_array_equal(mask.cpu().numpy(), z["masked_index"])
        np.testing.assert_allclose(logits_AE.cpu().numpy(), z["logits_AE"], **tol)
        np.testing.assert_allclose(float(ce), float(z["ce_mask_loss"]), rtol=1e-4 if prec == "f32" else 5e-3)


def _check_grads(model, z, tag):
    worst = []
    for name, p in model.named_parameters():
        key = tag + ":" + name
        if key not in z.files:
            assert p.grad is None or float(p.grad.abs().max()) == 0.0, name      # (the branch that was not part of the loss)
            continue
        ref, got = z[key].astype(np.float32), p.grad.detach().float().cpu().numpy()
        err, rn = np.linalg.norm(got - ref), np.linalg.norm(ref)
        worst.append((err / max(rn, 1e-12), err, name))
    bad = [(r, e, n) for r, e, n in worst if r >= 6e-2 and e >= 5e-3]              # the bf16 gradient bound of tests/test_gpu_trainer.py
    assert not bad, bad
    assert np.median([w[0] for w in worst]) < 2.5e-2


def test_pretraining_step_gradients(golden_dir):
    z, model = build(golden_dir)
    ids, lens = torch.from_numpy(z["ids"]).to(DEV), torch.from_numpy(z["lens"]).to(DEV)
    asr_amd.set_precision("bf16")
    logits_AE, _, mask = model(ids, lens, rand=rand_of_fixture(z["ids"].shape).to(DEV))
    ce = mask_lm.cal_ce_mask_loss(logits_AE, ids, mask, smoothing=0.1)            # solver.py:95-97
    model.zero_grad()
    ce.backward()
    np.testing.assert_allclose(float(ce), float(z["ce_mask_loss"]), rtol=5e-3)
    _check_grads(model, z, "gpre")


def test_finetuning_step_gradients(golden_dir):
    z, model = build(golden_dir)
    ids, lens, ys = (torch.from_numpy(z[k]).to(DEV) for k in ("ids", "lens", "ys"))
    asr_amd.set_precision("bf16")
    _, logits, none = model(ids, lens, padded_target=ys, mask_input=False)          # solver.py:228-230
    assert none is None
    np.testing.assert_allclose(logits.detach().cpu().numpy(), z["logits"], atol=6e-2, rtol=2e-2)
    ctc = mask_lm.cal_ctc_loss(logits, lens, ys)
    model.zero_grad()
    ctc.backward()
    np.testing.assert_allclose(float(ctc), float(z["ctc_loss"]), rtol=5e-3)
    _check_grads(model, z, "gft")
