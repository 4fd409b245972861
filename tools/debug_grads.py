import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import asr_amd
from weights import make_state_dict, names_shapes_from_json
z = np.load(os.path.join(ROOT, "tests/golden/g1_ctc_transformer.npz"))
sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to("cuda:0").eval()
tr = asr_amd.Trainer(model)
x, lens, tg = (torch.from_numpy(z[k]).to("cuda:0") for k in ("x", "lens", "targets"))
tr.fp.grad.zero_()
ctc, ce, st = tr.forward_loss(x, lens, tg)
tr.backward(st)
torch.cuda.synchronize()
print("ctc", float(ctc), z["ctc_loss"], "ce", float(ce), z["ce_loss_s01"])
for name, p in model.named_parameters():
    ref = z["grad:" + name].astype(np.float32); got = p.grad.float().cpu().numpy()
    print("%-50s ref_norm %.3e got_norm %.3e rel %.3e" % (name, np.linalg.norm(ref), np.linalg.norm(got), np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)))
