"""Per-op time table of one training step (live HIP-event timing of every C-ABI call): python tools/step_breakdown.py [--dropout p]"""
import os, sys, json, argparse
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import bench
import asr_amd
from asr_amd import ops
ap = argparse.ArgumentParser(); ap.add_argument("--dropout", type=float, default=0.1); ap.add_argument("--steps", type=int, default=5)
a = ap.parse_args()
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, a.dropout, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
for _ in range(3): tr.step(x, lens, tg)
torch.cuda.synchronize()
ops.profile_start()
for _ in range(a.steps): tr.step(x, lens, tg)
prof = ops.profile_stop()
rows = sorted(((r["ms"] / a.steps, n, r["calls"] / a.steps) for n, r in prof.items()), reverse=True)
tot = sum(r[0] for r in rows)
fam = {}
for ms, n, c in rows:
    f = n.split("[")[0]
    fam[f] = fam.get(f, 0) + ms
print("total timed %.2f ms/step" % tot)
for f, ms in sorted(fam.items(), key=lambda t: -t[1]): print("  %-22s %6.3f ms  %4.1f%%" % (f, ms, 100 * ms / tot))
print()
for ms, n, c in rows[:70]: print("%-44s %6.3f ms/step  %5.1f calls  %7.1f us/call" % (n, ms, c, ms / c * 1e3))
