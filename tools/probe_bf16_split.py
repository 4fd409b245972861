"""does a two-term bf16 split at the vocabulary projections bring the bf16 path's S0 logits to 3e-2?

Runs fixture G0 (Conv_CTC_Transformer, S0) on the GPU in bf16 mode, captures the fp32 master and the bf16 shadow of the two
projections' inputs, and re-evaluates the projections four ways (fp32 accumulation of bf16-exact products, as the MFMA does):
  a  x_hi . W_hi                       (what the product path computes)
  b  (x_hi + x_lo) . W_hi              (input split: 2 terms)
  c  (x_hi + x_lo) . (W_hi + W_lo)     (input and weight split, x_lo . W_lo dropped: 3 terms)
  d  x_f32 . W_f32                     (no rounding at the projection at all)
against the reference's logits.  Whatever error (d) keeps is upstream of the projection."""
import argparse
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import asr_amd
from asr_amd import modules
from weights import make_state_dict, names_shapes_from_json

z = np.load(os.path.join(ROOT, "tests", "golden", "g0_conv_ctc_transformer.npz"))
sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
dev = torch.device("cuda:0")
model = asr_amd.Conv_CTC_Transformer.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to(dev).eval()
x, lens, tg = (torch.from_numpy(z[k]).to(dev) for k in ("x", "lens", "targets"))

seen = {}
orig = modules._vocab_proj


def spy(mod, key, weight, xa, want_lse=False):
    seen[key] = (xa.f32.clone(), None if xa.b16 is None else xa.b16.clone(), weight.detach().clone())
    return orig(mod, key, weight, xa, want_lse)


modules._vocab_proj = spy
with asr_amd.precision("bf16"), torch.no_grad():
    out = model(x, lens, tg)
modules._vocab_proj = orig
torch.cuda.synchronize()
got = {"ctc": out[0].float().cpu().numpy(), "prj": out[2].float().cpu().numpy()}
ref = {"ctc": z["ctc_logits"], "prj": z["logits"]}
for key in ("ctc", "prj"):
    x32, x16, w = seen[key]
    r = ref[key].reshape(-1, ref[key].shape[-1])
    valid = np.abs(r).sum(1) > 0          # padded encoder rows give exact-zero logits on both sides
    xh = (x16 if x16 is not None else x32.bfloat16()).float()
    xl = (x32 - xh).bfloat16().float()
    wh = w.bfloat16().float()
    wl = (w - wh).bfloat16().float()
    mm = lambda a, b: (a.double() @ b.double().t()).float().cpu().numpy()
    variants = {"a x_hi.W_hi": mm(xh, wh), "b +x_lo.W_hi": mm(xh, wh) + mm(xl, wh), "c +x_hi.W_lo": mm(xh, wh) + mm(xl, wh) + mm(xh, wl),
                "d f32.f32": mm(x32, w)}
    print("%s  (max |ref| %.3f; the product path itself: max err %.4e)" % (key, np.abs(r).max(), np.abs(got[key].reshape(r.shape) - r).max()))
    for name, v in variants.items():
        e = np.abs(v[: r.shape[0]] - r)[valid]
        print("   %-14s max %.4e   rel-L2 %.4e" % (name, e.max(), np.linalg.norm(v[: r.shape[0]][valid] - r[valid]) / np.linalg.norm(r[valid])))
