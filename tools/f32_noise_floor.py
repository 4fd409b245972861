#!/usr/bin/env python3
"""How far apart are two fp32 evaluations of the S1 gradients?  torch CPU f64 as the arbiter: error of the GPU fp32 parity mode and
of torch CPU fp32 against it, per parameter, on 2 ragged utterances at full model size."""
import json
import os
import sys

R0 = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, R0)
import numpy as np
import torch

import asr_amd
import bench
from oracle import torch_cpu_ref as R

dev = torch.device("cuda:0")
bench.CFG["n_conv_layers"] = int(os.environ.get("CONV", "0"))
model = bench.build_model(asr_amd, dev, 0.0, train=True)
x, lens, tg = bench.make_batch(dev, seed=0, ragged=True)
x, lens, tg = x[:2].contiguous(), lens[:2].clone(), tg[:2].contiguous()
T = bench.CFG["T"]
lens[:] = torch.tensor([T, T - 137], device=dev)
x[1, T - 137:] = 0
cfg = dict(n_head=4, n_layers_enc=12, n_layers_dec=6, sos_id=bench.CFG["sos_id"], eos_id=bench.CFG["eos_id"])
torch.set_num_threads(min(32, os.cpu_count()))


def cpu(dtype):
    sd = {k: (v.detach().cpu().to(dtype).requires_grad_(not k.endswith(".pe"))) for k, v in model.state_dict().items()}
    ctc, ce, _, _ = R.joint_step(sd, x.cpu().to(dtype), lens.cpu(), tg.cpu(), cfg, conv_layers=bench.CFG["n_conv_layers"], p=0.0, train=False,
                                 smoothing=0.1, backward=True)
    return float(ctc), float(ce), {k: v.grad.double().numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}


c64 = cpu(torch.float64)
c32 = cpu(torch.float32)
with asr_amd.precision("f32"):
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    tr.fp.grad.zero_()
    ctc, ce, state = tr.forward_loss(x, lens, tg)
    tr.backward(state)
    torch.cuda.synchronize()
    g32 = {n: p.grad.detach().double().cpu().numpy() for n, p in model.named_parameters()}
rows = []
for n in g32:
    r = c64[2][n]
    rn = np.linalg.norm(r)
    rows.append((np.linalg.norm(g32[n] - r) / max(rn, 1e-30), np.linalg.norm(c32[2][n] - r) / max(rn, 1e-30), rn, n))
rows.sort(reverse=True)
print(json.dumps(dict(loss64=c64[:2], loss_cpu32=c32[:2], loss_gpu32=(float(ctc), float(ce)))))
for a, b, rn, n in rows[:12]:
    print("gpu32-vs-64 %.2e   cpu32-vs-64 %.2e   |g| %.3e  %s" % (a, b, rn, n))
ga = np.sqrt(sum((np.linalg.norm(g32[n] - c64[2][n])) ** 2 for n in g32)) / np.sqrt(sum(np.linalg.norm(c64[2][n]) ** 2 for n in g32))
ca = np.sqrt(sum((np.linalg.norm(c32[2][n] - c64[2][n])) ** 2 for n in g32)) / np.sqrt(sum(np.linalg.norm(c64[2][n]) ** 2 for n in g32))
print("whole vector: gpu32 %.2e  cpu32 %.2e" % (ga, ca))
