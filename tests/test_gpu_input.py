"""LFR stacking of a padded batch and SpecAugment on the device against the reference's outputs (tests/golden/g10_input.npz:
utils/data.py:191-218, utils/utils.py:168-194), and SpecAugment inside a model's forward."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import data

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def test_lfr_batch_matches_per_utterance_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g10_input.npz"))
    Ts = [100, 9, 8, 7, 1]
    xs = [z["lfr_x_T%d" % T] for T in Ts]
    pad = np.zeros((len(Ts), 100, 6), np.float32)
    for i, x in enumerate(xs):
        pad[i, :len(x)] = x
    for m, n in ((4, 3), (1, 2), (3, 1), (1, 1), (7, 6)):
        y, l = data.lfr_batch(torch.from_numpy(pad).to(DEV), torch.tensor(Ts).to(DEV), m, n)
        y, l = y.cpu().numpy(), l.cpu().numpy()
        for i, T in enumerate(Ts):
            ref = z["lfr_T%d_m%d_n%d" % (T, m, n)]
            assert l[i] == len(ref)
            np.testing.assert_array_equal(y[i, :len(ref)], ref)
            assert not y[i, len(ref):].any()                       # zero padding beyond the stacked length


@pytest.mark.parametrize("cfg", ["2-5-2-8", "1-16-3-12", "2-3-1-4"])
def test_spec_aug_matches_reference_under_the_same_draws(golden_dir, cfg):
    z = np.load(os.path.join(golden_dir, "g10_input.npz"))
    x = torch.from_numpy(z["sa_x"]).to(DEV)
    lens = torch.from_numpy(z["sa_lens"]).to(DEV)
    torch.manual_seed(1010)                                        # the reference drew torch.rand(size=[B]) on the CPU from this seed
    rand = data.spec_aug_draws(cfg, x.shape[0], "cpu")
    y, l = data.spec_aug(x, lens, cfg, rand=rand)
    assert y.data_ptr() == x.data_ptr()                            # in place, like the reference
    ref = z["sa_y_" + cfg]
    assert (ref != z["sa_x"]).mean() > 0.02                        # the fixture does mask something
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-6, atol=2e-7)
    changed = ref != z["sa_x"]
    np.testing.assert_array_equal(y.cpu().numpy()[~changed], z["sa_x"][~changed])


def test_model_forward_applies_spec_aug(golden_dir):
    torch.manual_seed(0)
    enc = asr_amd.Encoder(80, 1, 2, 64, 128, dropout=0.0)
    dec = asr_amd.Decoder(2, 3, 50, 1, 2, 64, 128, dropout=0.0)
    model = asr_amd.CTC_Transformer(enc, dec, spec_aug_cfg="2-27-2-40").to(DEV).eval()
    x = torch.randn(3, 120, 80, device=DEV)
    lens = torch.tensor([120, 100, 90], device=DEV)
    x = x * (torch.arange(120, device=DEV)[None, :, None] < lens[:, None, None])
    before = x.clone()
    tg = torch.randint(4, 49, (3, 6), device=DEV)
    with torch.no_grad():
        l, ctc_logits, (logits, teos) = model(x, lens, tg)
    assert bool(torch.isfinite(ctc_logits).all())
    assert float((x != before).float().mean()) > 0.05              # the batch was masked in place (transformer.py:116-117)
