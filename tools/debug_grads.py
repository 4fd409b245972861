"""Per-parameter gradient error of the HIP training step against the reference's gradients in a golden fixture.
usage: python tools/debug_grads.py [g1|g6|g4|g7]"""
import argparse
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import asr_amd
from weights import make_state_dict, names_shapes_from_json

which = sys.argv[1] if len(sys.argv) > 1 else "g1"
fn = {"g1": "g1_ctc_transformer.npz", "g6": "g6_ctc_transformer_train.npz", "g4": "g4_cif_model.npz", "g7": "g7_cif_model_train.npz"}[which]
z = np.load(os.path.join(ROOT, "tests/golden", fn))
sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
p = float(z["drop_p"]) if "drop_p" in z.files else 0.0
if which in ("g1", "g6"):
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=p), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=p))
    kw = {}
else:
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    cfg["dropout"] = p
    model = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    kw = {"lambda_qua": 0.001}
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to("cuda:0")
model = model.train() if p > 0 else model.eval()
if p > 0:
    asr_amd.manual_seed(int(z["drop_seed"]))
tr = asr_amd.Trainer(model, **kw)
x, lens, tg = (torch.from_numpy(z[k]).to("cuda:0") for k in ("x", "lens", "targets"))
noise = torch.from_numpy(z["noise"]).to("cuda:0") if "noise" in z.files else None
tr.fp.grad.zero_()
ctc, ce, st = tr.forward_loss(x, lens, tg, noise=noise)
tr.backward(st)
torch.cuda.synchronize()
print("ctc", float(ctc), z["ctc_loss"], "ce", float(ce), z["ce_loss_s01"])
for name, q in model.named_parameters():
    ref = z["grad:" + name].astype(np.float32)
    got = q.grad.float().cpu().numpy()
    print("%-50s ref_norm %.3e got_norm %.3e rel %.3e" % (name, np.linalg.norm(ref), np.linalg.norm(got), np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)))
