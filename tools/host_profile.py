"""Where the HOST spends its time queueing one S1 training step (cProfile over eager steps; the GPU runs behind): python tools/host_profile.py"""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import asr_amd
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, 0.1, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(5):
    tr.step(x, lens, tg, max_target_len=50)
torch.cuda.synchronize()
N = 10
t0 = time.perf_counter()
for _ in range(N):
    tr.step(x, lens, tg, max_target_len=50)
t_q = (time.perf_counter() - t0) / N * 1e3
torch.cuda.synchronize()
t_all = (time.perf_counter() - t0) / N * 1e3
print("host queueing %.2f ms/step, with the GPU drained %.2f ms/step" % (t_q, t_all))
pr = cProfile.Profile()
pr.enable()
for _ in range(N):
    tr.step(x, lens, tg, max_target_len=50)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("tottime").print_stats(28)
print(s.getvalue()[:6000])
