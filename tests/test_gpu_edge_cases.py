"""Edge batches through the whole step (forward, joint loss, backward) against the stock-torch restatement of the reference's op
sequence (oracle/torch_cpu_ref.py, pinned on the reference's outputs and gradients): an utterance with an EMPTY target (all pad:
decoder input = <sos> only, CTC of the empty label sequence), a one-token target, very short inputs beside a full-length one, and
a batch of one."""
import numpy as np
import pytest
import torch

import asr_amd
from oracle import torch_cpu_ref as R

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
CFG = dict(n_head=2, n_layers_enc=2, n_layers_dec=2, sos_id=2, eos_id=3)
V = 50


def build(conv):
    torch.manual_seed(11 + conv)
    enc = asr_amd.Encoder(64 if conv else 80, 2, 2, 64, 128, dropout=0.0)
    dec = asr_amd.Decoder(2, 3, V, 2, 2, 64, 128, dropout=0.0)
    if conv:
        return asr_amd.Conv_CTC_Transformer(asr_amd.Conv2dSubsample(80, 64, n_layers=2), enc, dec).to(DEV).eval()
    return asr_amd.CTC_Transformer(enc, dec).to(DEV).eval()


def batches():
    g = torch.Generator().manual_seed(5)
    T = 48
    x = torch.randn(4, T, 80, generator=g)
    lens = torch.tensor([T, 3, 17, 1])
    for b in range(4):
        x[b, lens[b]:] = 0
    tg = torch.zeros(4, 6, dtype=torch.long)
    tg[0, :6] = torch.randint(4, V - 1, (6,), generator=g)
    tg[1, :1] = 7                      # one token
    # row 2: empty target; row 3: one token on a one-frame input
    tg[3, :1] = 9
    yield "mixed-infeasible", x, lens, tg                      # row 3: one frame for two CTC labels (token + <eos>): the loss is inf
    lens2 = torch.tensor([T, 3, 17, 2])
    x2 = x.clone()
    x2[3, 1] = torch.randn(80, generator=g)
    yield "mixed", x2, lens2, tg
    yield "single", x[:1], lens[:1], tg[:1]


@pytest.mark.parametrize("conv", [0, 2])
def test_edge_batches_match_stock_torch(conv):
    model = build(conv)
    for name, x, lens, tg in batches():
        if conv and name.startswith("mixed"):
            # (lengths that subsample - ceil(len / 4) - to enough frames for the rows' CTC labels; "infeasible": 1 frame for 2 labels)
            lens = torch.tensor([48, 9, 17, 5 if name == "mixed" else 4])
        sd = {k: v.detach().cpu().double().requires_grad_(not k.endswith(".pe")) for k, v in model.state_dict().items()}
        ctc_ref, ce_ref, ctc_logits_ref, logits_ref = R.joint_step(sd, x.double(), lens, tg, CFG, conv_layers=conv, p=0.0, train=False,
                                                                   smoothing=0.1, backward=True)
        with asr_amd.precision("f32"):
            tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
            tr.fp.grad.zero_()
            ctc, ce, state = tr.forward_loss(x.to(DEV), lens.to(DEV), tg.to(DEV))
            tr.backward(state)
            torch.cuda.synchronize()
        assert np.isfinite(float(ctc_ref.detach())) == np.isfinite(float(ctc)), (name, float(ctc), float(ctc_ref.detach()))
        if np.isfinite(float(ctc)):
            np.testing.assert_allclose(float(ctc), float(ctc_ref.detach()), rtol=2e-5, err_msg=name)
        np.testing.assert_allclose(float(ce), float(ce_ref.detach()), rtol=2e-5, err_msg=name)
        if not np.isfinite(float(ctc)):
            continue
        for pname, p in model.named_parameters():
            ref = sd[pname].grad
            if ref is None:
                continue
            r, got = ref.numpy(), p.grad.detach().double().cpu().numpy()
            err, rn = np.linalg.norm(got - r), np.linalg.norm(r)
            assert err <= 5e-4 * rn + 1e-6, (name, pname, err, rn)
