"""Training step of Conv_CTC_Transformer at the S2 dimensions (conv front end: T = 1000 -> L = 250), B = 32, U = 50."""
import os, sys, argparse
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, asr_amd
B, T, U, V = 32, 1000, 50, 4234
args = argparse.Namespace(d_input=80, LFR_m=1, d_model=256, n_conv_layers=2, n_layers_enc=12, n_head=4, d_inner=2048, dropout=0.1,
                          sos_id=2, eos_id=3, vocab_size=V, n_layers_dec=6, spec_aug_cfg=None)
dev = torch.device("cuda", 0)
torch.manual_seed(0)
model = asr_amd.Conv_CTC_Transformer.create_model(args).to(dev).train()
asr_amd.manual_seed(1)
x = torch.randn(B, T, 80, device=dev); lens = torch.full((B,), T, device=dev, dtype=torch.int64); tg = torch.randint(4, V - 2, (B, U), device=dev)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(6): out = tr.step(x, lens, tg, max_target_len=U)
torch.cuda.synchronize()
a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
a.record()
for _ in range(20): out = tr.step(x, lens, tg, max_target_len=U)
b.record(); torch.cuda.synchronize()
ms = a.elapsed_time(b) / 20
print("Conv_CTC_Transformer S2 train step: %.3f ms  (%.2f M frames/s)  losses %s" % (ms, B * T / ms / 1e3, [round(float(v), 3) for v in out[:2]]))
