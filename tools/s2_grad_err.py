"""Which parameters carry the bf16 step's gradient error at the S2 shape (2 ragged utterances, float64 CPU autograd as the arbiter:
tests/test_gpu_fullsize.py::test_full_size_gradients...)?  Prints the whole-vector error and the parameters' shares of its square.
usage: python tools/s2_grad_err.py [s1|s2] [CONV_F32=1 ...module switches as NAME=value]"""
import os
import sys
import numpy as np
import torch
ROOT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
sys.path.insert(0, ROOT)
import asr_amd
import bench
from asr_amd import modules, ops
from oracle import torch_cpu_ref as R

which = sys.argv[1] if len(sys.argv) > 1 else "s2"
for kv in sys.argv[2:]:
    k, val = kv.split("=")
    setattr(modules, k, int(val))
DEV = "cuda:0"
bench.CFG["n_conv_layers"] = 2 if which == "s2" else 0
dev = torch.device(DEV)
model = bench.build_model(asr_amd, dev, 0.0, train=True)
x, lens, tg = bench.make_batch(dev, seed=0, ragged=True)
NU = int(os.environ.get("NUTT", "2"))
x, lens, tg = x[:NU].contiguous(), lens[:NU].clone(), tg[:NU].contiguous()
T = bench.CFG["T"]
lens[:2] = torch.tensor([T, T - 137], device=DEV)
x[1, T - 137:] = 0
cfg = dict(n_head=bench.CFG["n_head"], n_layers_enc=bench.CFG["n_layers_enc"], n_layers_dec=bench.CFG["n_layers_dec"],
           sos_id=bench.CFG["sos_id"], eos_id=bench.CFG["eos_id"])
sd = {k: v.detach().cpu().double().requires_grad_(not k.endswith(".pe")) for k, v in model.state_dict().items()}
torch.set_num_threads(max(1, min(32, os.cpu_count() or 1)))
R.joint_step(sd, x.cpu().double(), lens.cpu(), tg.cpu(), cfg, conv_layers=bench.CFG["n_conv_layers"], p=0.0, train=False, smoothing=0.1, backward=True)
ref = {k: v.grad.double().numpy() for k, v in sd.items() if v.requires_grad and v.grad is not None}
for prec in ("bf16",):
    with asr_amd.precision(prec):
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
        tr.fp.grad.zero_()
        ctc, ce, state = tr.forward_loss(x, lens, tg)
        tr.backward(state)
        torch.cuda.synchronize()
        errs = []
        for name, p in model.named_parameters():
            g, r = p.grad.detach().double().cpu().numpy(), ref[name]
            errs.append((float(np.linalg.norm(g - r)), float(np.linalg.norm(r)), name))
    tot_e, tot_r = np.sqrt(sum(e * e for e, r, n in errs)), np.sqrt(sum(r * r for e, r, n in errs))
    print("%s %s %d utterances: whole vector %.3e (|g| %.3e)" % (which, prec, NU, tot_e / tot_r, tot_r))
    for e, r, n in sorted(errs, key=lambda t: -t[0])[:14]:
        print("  %-55s share of err^2 %5.1f %%   rel %.2e   |g| share %5.1f %%" % (n, 100 * e * e / (tot_e * tot_e), e / max(r, 1e-30), 100 * r * r / (tot_r * tot_r)))
