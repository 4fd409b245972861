"""Where does the bf16 path lose accuracy at S0 (tests/golden g0)?  Per-module max |bf16 - f32| of the module outputs (both on the GPU)."""
import os, sys
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import argparse, numpy as np, torch
import asr_amd
from weights import make_state_dict, names_shapes_from_json
z = np.load(os.path.join(ROOT, "tests", "golden", "g0_conv_ctc_transformer.npz"))
sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
dev = torch.device("cuda:0")
model = asr_amd.Conv_CTC_Transformer.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
model = model.to(dev).eval()
x, lens, tg = (torch.from_numpy(z[k]).to(dev) for k in ("x", "lens", "targets"))
outs = {}
def hook(name):
    def f(m, i, o):
        t = o[0] if isinstance(o, (tuple, list)) else o
        if torch.is_tensor(t):
            outs.setdefault(name, []).append(t.detach().float().cpu())
    return f
for n, m in model.named_modules():
    if n and n.count(".") <= 2:
        m.register_forward_hook(hook(n))
res = {}
for prec in ("f32", "bf16"):
    outs.clear()
    with asr_amd.precision(prec), torch.no_grad():
        r = model(x, lens, tg)
    res[prec] = {k: v[0] for k, v in outs.items()}
    res[prec]["__ctc"] = r[0].float().cpu(); res[prec]["__logits"] = r[2].float().cpu()
for k in res["f32"]:
    a, b = res["f32"][k], res["bf16"].get(k)
    if b is not None and a.shape == b.shape:
        print("%-44s max|f32| %8.3f   max|bf16 - f32| %.4f   rel %.4f" % (k, float(a.abs().max()), float((a - b).abs().max()), float((a - b).norm() / a.norm())))
