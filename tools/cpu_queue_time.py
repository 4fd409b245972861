"""How long does the host take to QUEUE one training step (no sync inside) vs the GPU to run it?"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import asr_amd
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, 0.1, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(3): tr.step(x, lens, tg)
torch.cuda.synchronize()
qs, gs = [], []
for _ in range(5):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    tr.fp.grad.zero_()
    ctc, ce, st = tr.forward_loss(x, lens, tg)
    t1 = time.perf_counter()
    tr.backward(st)
    t2 = time.perf_counter()
    tr.optimizer_step()
    t3 = time.perf_counter()
    torch.cuda.synchronize()
    t4 = time.perf_counter()
    qs.append((t1 - t0, t2 - t1, t3 - t2)); gs.append(t4 - t0)
import statistics as S
print("host queue time: forward+loss %.2f ms, backward %.2f ms, adam %.2f ms | step wall (queue+drain) %.2f ms" % (
    1e3 * S.median(q[0] for q in qs), 1e3 * S.median(q[1] for q in qs), 1e3 * S.median(q[2] for q in qs), 1e3 * S.median(gs)))
