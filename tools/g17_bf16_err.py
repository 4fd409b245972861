"""bf16 gradient errors of the plain Transformer on fixture G17, per parameter (trainer path and the autograd bridge)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import torch

import asr_amd
import test_gpu_transformer_plain as T

gd = os.path.join(ROOT, "tests", "golden")
for fixture, build in (("g17", T.build),):
    z, sd, model = build(gd)
    x, lens, tg = T.batch(z)
    asr_amd.set_precision("bf16")
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    torch.cuda.synchronize()
    rows = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.detach().float().cpu().numpy()
        rows.append((float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)), float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref)), name))
    rows.sort(reverse=True)
    print("trainer path: median rel %.3e" % np.median([r[0] for r in rows]))
    for r in rows[:12]:
        print("   rel %.3e  err %.3e  |ref| %.3e  %s" % r)
    z, sd, model = build(gd)
    model.zero_grad()
    logits, teos = model(x, lens, tg)
    asr_amd.cal_ce_loss(logits, teos, smoothing=0.1).backward()
    torch.cuda.synchronize()
    rows = []
    for name, p in model.named_parameters():
        ref = z["grad:" + name].astype(np.float32)
        got = p.grad.detach().float().cpu().numpy()
        rows.append((float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)), float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref)), name))
    rows.sort(reverse=True)
    print("autograd bridge: median rel %.3e" % np.median([r[0] for r in rows]))
    for r in rows[:12]:
        print("   rel %.3e  err %.3e  |ref| %.3e  %s" % r)
# the same statistic for G1 (CTC_Transformer), for scale
sys.path.insert(0, os.path.join(ROOT, "tests"))
import test_gpu_trainer as TT
z, sd, model = TT.build(gd)
asr_amd.set_precision("bf16")
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
x, lens, tg = (torch.from_numpy(z[k]).to("cuda:0") for k in ("x", "lens", "targets"))
tr.fp.grad.zero_()
ctc, ce, state = tr.forward_loss(x, lens, tg)
tr.backward(state)
torch.cuda.synchronize()
rows = []
for name, p in model.named_parameters():
    ref = z["grad:" + name].astype(np.float32)
    got = p.grad.detach().float().cpu().numpy()
    rows.append((float(np.linalg.norm(got - ref) / max(np.linalg.norm(ref), 1e-12)), float(np.linalg.norm(got - ref)), float(np.linalg.norm(ref)), name))
rows.sort(reverse=True)
print("G1 trainer path: median rel %.3e" % np.median([r[0] for r in rows]))
for r in rows[:8]:
    print("   rel %.3e  err %.3e  |ref| %.3e  %s" % r)
