#!/usr/bin/env python3
"""Soak of the replayed step WITH its collective nodes (a 1-rank RCCL communicator owned by libasr_hip.so): N steps through
Trainer.step_graphed(force_collective), watching memory, the losses and the communicator's asynchronous error state."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import asr_amd
import bench
from asr_amd import ops

dev = torch.device("cuda:0")
N = int(os.environ.get("STEPS", "1000"))
model = bench.build_model(asr_amd, dev, 0.1, train=True)
asr_amd.manual_seed(7)
x, lens, tg = bench.make_batch(dev, seed=0, ragged=True)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1, force_collective=True)
rows = []
for i in range(N):
    ctc, ce = tr.step_graphed(x, lens, tg, max_target_len=bench.CFG["U"])
    if i % (N // 8) == 0 or i == N - 1:
        torch.cuda.synchronize()
        ops.rccl_comm().check()
        rows.append((i, float(ctc), float(ce), torch.cuda.memory_allocated() / 2**20))
assert tr.graph_active() and tr._graphx is not None and tr._graphx.info["collectives"] == len(tr.buckets.ranges) + 1, tr._graph_failed
for r in rows:
    print("step %4d  ctc %9.3f  ce %7.4f  allocated %8.1f MiB" % r)
assert rows[-1][3] <= rows[2][3] * 1.02 + 64 and all(r[1] == r[1] and r[2] == r[2] for r in rows)
print("soak with collective nodes ok:", tr._graphx.info)
