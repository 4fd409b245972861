"""Per-op time table of one CIF_Model training step at the S2 dimensions (conv front end, L = T/4): python tools/cif_step_breakdown.py"""
import os, sys, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import asr_amd
from asr_amd import ops
B, T, U, V = 32, 1000, 50, 4234
args = argparse.Namespace(d_input=80, LFR_m=1, d_model=256, n_conv_layers=2, n_layers_enc=12, n_head=4, d_inner=2048, dropout=0.1,
                          sos_id=2, eos_id=3, vocab_size=V, n_layers_dec=6, spec_aug_cfg=None, d_assigner_hidden=256, w_context=3,
                          n_assigner_layers=3)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = asr_amd.CIF_Model.create_model(args).to(dev).train()
asr_amd.manual_seed(1)
x = torch.randn(B, T, 80, device=dev)
lens = torch.full((B,), T, device=dev, dtype=torch.int64)
tg = torch.randint(4, V - 2, (B, U), device=dev)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(3): out = tr.step(x, lens, tg)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(10): out = tr.step(x, lens, tg)
b.record(); torch.cuda.synchronize()
print("CIF_Model S2 train step: %.3f ms  (%.2f M frames/s)  losses %s" % (a.elapsed_time(b) / 10, B * T / (a.elapsed_time(b) / 10) / 1e3, [float(v) for v in out[:3]]))
out = tr.step_auto(x, lens, tg, max_target_len=U)          # no host read-back in the step; eager vs hipGraph replay, calibrated
torch.cuda.synchronize()
a.record()
for _ in range(10): out = tr.step_auto(x, lens, tg, max_target_len=U)
b.record(); torch.cuda.synchronize()
print("  with max_target_len, step_auto -> %s %s: %.3f ms  (%.2f M frames/s)  losses %s" % (tr.launch_mode, tr.launch_timing, a.elapsed_time(b) / 10, B * T / (a.elapsed_time(b) / 10) / 1e3, [float(v) for v in out[:3]]))
tr._graph = None
ops.profile_start()
for _ in range(5): tr.step(x, lens, tg)
prof = ops.profile_stop()
rows = sorted(((r["ms"] / 5, n, r["calls"] / 5) for n, r in prof.items()), reverse=True)
print("timed ops total %.2f ms/step" % sum(r[0] for r in rows))
for ms, n, c in rows[:25]: print("%-44s %6.3f ms/step  %5.1f calls  %7.1f us/call" % (n, ms, c, ms / c * 1e3))
