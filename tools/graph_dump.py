"""Dump the captured S1 training step's hipGraph as DOT (hipGraphDebugDotPrint) and list the non-kernel nodes with their neighbours."""
import os, re, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
os.environ["ASR_AMD_GRAPH_EXEC"] = "0"
import torch
import bench
import asr_amd
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, 0.1, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
import torch.cuda
orig = torch.cuda.CUDAGraph
for _ in range(4):
    tr.step_graphed(x, lens, tg, max_target_len=50)
torch.cuda.synchronize()
g = tr._graph
g.enable_debug_mode() if False else None
path = os.path.join(ROOT, "gpurun_out", "step_graph.dot")
try:
    g.debug_dump(path)
except Exception as e:
    print("debug_dump failed:", e)
if os.path.exists(path):
    txt = open(path).read()
    print(len(txt), "bytes")
    for ln in txt.splitlines():
        if re.search(r"MEMCPY|MEMSET|memcpy|memset|EMPTY|empty", ln):
            print(ln[:300])
