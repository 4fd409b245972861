"""Where do device-to-device copies come from in a training step?  torch.profiler memcpy events with the launching op / stack."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, asr_amd, bench
from torch.profiler import profile, ProfilerActivity
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, 0.1, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(3): tr.step(x, lens, tg, max_target_len=50)
torch.cuda.synchronize()
with profile(activities=[ProfilerActivity.CPU, ProfilerActivity.CUDA], with_stack=True) as prof:
    tr.step(x, lens, tg, max_target_len=50)
    torch.cuda.synchronize()
cnt = collections.Counter()
for e in prof.events():
    n = e.name.lower()
    if "memcpy" in n or "copybuffer" in n or "memset" in n:
        st = [s for s in (e.stack or []) if "asr" in s][:2]
        cnt[(e.name, tuple(st))] += 1
for (n, st), c in cnt.most_common(30): print(c, n, st)
print("--- top cpu ops launching copies")
for e in prof.key_averages(group_by_stack_n=4).table(sort_by="count", row_limit=0).splitlines()[:0]: print(e)
names = collections.Counter(e.name for e in prof.events() if e.device_type is not None and "copy" in e.name.lower())
print(names.most_common(10))
