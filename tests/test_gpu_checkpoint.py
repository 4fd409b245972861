"""The reference's checkpoint package into the HIP trainer: weights + Adam moments + step counter from
tests/golden/g8_checkpoint.npz (the reference's serialize() after two optimizer steps), then one more step on the MI355X lands
where the reference's third step did."""
import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import checkpoint
from test_checkpoint import build, load_fixture, with_pe

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_resume_from_reference_package_and_take_its_third_step(golden_dir):
    z, cfg, package, keys, pe_keys = load_fixture(golden_dir)
    model = build(cfg).to(DEV).train()
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    checkpoint.load_package(with_pe(package, model, pe_keys), model, tr)
    assert tr.step_num == 2
    # loaded in place: the parameters still live in the trainer's flat buffer, and the bf16 shadow the kernels read follows them
    p = dict(model.named_parameters())["ctc_fc.weight"]
    assert p.data_ptr() == tr.fp.flat[p._asr_off:].data_ptr()
    np.testing.assert_array_equal(p.detach().cpu().numpy(), z["sd:ctc_fc.weight"])
    np.testing.assert_array_equal(tr.fp.flat16.float().cpu().numpy(), tr.fp.flat.bfloat16().float().cpu().numpy())
    before = {k: q.detach().clone() for k, q in model.named_parameters()}
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    ctc, ce = tr.step(x, lens, tg)
    torch.cuda.synchronize()
    assert tr.step_num == 3
    np.testing.assert_allclose(tr.lr(), float(z["lr_step3"]), rtol=1e-12)
    np.testing.assert_allclose(float(ctc) + float(ce), float(z["loss3"]), rtol=5e-3)
    # third Adam step: m / v carry two steps of the reference's history, so the update is lr * mhat / (sqrt(vhat) + eps) of BOTH
    # histories - compare the parameter movement with the reference's, tensor by tensor (bf16 gradients: a few % of the step)
    names = [n for n in keys if not n.endswith(".pe")]
    lr, b1, b2, eps = tr.lr(), 0.9, 0.98, 1e-9
    num2 = den2 = own2 = 0.0
    for k, q in model.named_parameters():
        if k.endswith("w_ks.bias"):      # its gradient is zero analytically (softmax is shift-invariant per query): Adam normalises pure noise
            continue
        got = (q.detach() - before[k]).double().cpu().numpy()
        ref = (z["after3:" + k].astype(np.float64) - z["sd:" + k])
        # (i) the optimizer itself: torch.optim.Adam's third step from the PACKAGE's moments and THIS step's gradient
        i = names.index(k)
        g = q.grad.double().cpu().numpy()
        m = b1 * z["opt:%d:exp_avg" % i] + (1 - b1) * g
        v = b2 * z["opt:%d:exp_avg_sq" % i] + (1 - b2) * g * g
        exp = -lr * (m / (1 - b1 ** 3)) / (np.sqrt(v) / np.sqrt(1 - b2 ** 3) + eps)
        # p_after - p_before is only known to half an ulp of |p| on either side (lr = 4e-7 here: a few dozen ulps of a weight of 0.1)
        ulp = np.abs(before[k].double().cpu().numpy()) * 1.2e-7 + 1e-12
        assert np.all(np.abs(got - exp) <= ulp + 2e-3 * lr), k
        # (ii) against the reference's step: the bf16 gradient moves m / sqrt(v) of elements whose history is small
        num, den = np.linalg.norm(got - ref), np.linalg.norm(ref)
        own2 += np.linalg.norm(exp - ref) ** 2
        num2, den2 = num2 + num ** 2, den2 + den ** 2
    assert num2 ** 0.5 <= 0.15 * den2 ** 0.5, (num2 ** 0.5, den2 ** 0.5)      # all parameters together: within 15 % of the reference's step
    # and the package written back holds what the trainer holds
    out = checkpoint.serialize(model, tr, epoch=3)
    i = [n for n in keys if not n.endswith(".pe")].index("ctc_fc.weight")
    q = dict(model.named_parameters())["ctc_fc.weight"]
    np.testing.assert_array_equal(out["optim_dict"]["state"][i]["exp_avg"].numpy(),
                                  tr.m[q._asr_off:q._asr_off + q.numel()].view(q.shape).cpu().numpy())
    assert float(out["optim_dict"]["state"][i]["step"]) == 3.0
