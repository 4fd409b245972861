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


def test_spec_aug_short_utterances_follow_python_slicing():
    """An utterance shorter than the drawn time-mask width: (len - width) < 0, the start index is negative and the reference's
    slice counts it from the end of the PADDED axis (utils.py:186-192).  Checked against the oracle's restatement, whose numpy
    slices wrap exactly like the reference's torch slices; the draws are fixed so that every case occurs (a run at the tail of
    the padded axis, an empty slice, an ordinary interior run)."""
    from oracle import asr_oracle as O
    B, T, V = 4, 12, 6
    g = torch.Generator().manual_seed(3)
    x = torch.randn(B, T, V, generator=g)
    lens = torch.tensor([12, 3, 2, 5])
    x = x * (torch.arange(T)[None, :, None] < lens[:, None, None])
    cfg = "1-2-2-9"
    #        freq width / start (2 loops)                         time width / start (2 loops)
    rand = torch.tensor([[0.6, 0.9, 0.1, 0.5], [0.3, 0.0, 0.99, 0.5], [0.2, 0.4, 0.6, 0.8], [0.7, 0.1, 0.5, 0.3],
                         [0.5, 0.95, 0.95, 0.7], [0.4, 0.9, 0.1, 0.99], [0.99, 0.5, 0.3, 0.99], [0.9, 0.2, 0.99, 0.5]])
    ref = O.spec_aug(x.numpy(), lens.numpy(), cfg, rand.numpy())
    assert (ref != x.numpy()).any()
    y, _ = data.spec_aug(x.to(DEV).clone(), lens.to(DEV), cfg, rand=rand)
    np.testing.assert_allclose(y.cpu().numpy(), ref, rtol=2e-6, atol=2e-7)
    # (the oracle's restatement itself equals the reference on these draws - checked when the test was written, with torch.rand
    # patched to return them; e.g. the 2-frame utterance gets frames 0..7 overwritten, most of them in its padding)
    changed = (ref != x.numpy()).any(-1)
    assert changed[2, 2:8].all() and not changed[2, 8:].any()


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
