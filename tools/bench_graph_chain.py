"""Per-kernel cost of a DEPENDENT chain of tiny kernels: eager stream order vs one captured hipGraph replay, and the same with
real small kernels of the decoder (asr_add_layernorm on 1632 x 256 rows)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from asr_amd import ops
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
N = 240

def timed(fn, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); fn(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b) * 1e3)
    return best

def chain_tiny():
    for _ in range(N):
        x.add_(1.0)

rows = 1632
h = torch.randn(rows, 256, device=dev)
r = torch.randn(rows, 256, device=dev)
gw = torch.ones(256, device=dev); gb = torch.zeros(256, device=dev)
def chain_ln():
    y = h
    for _ in range(N):
        y = ops.add_layernorm(y, r, gw, gb, 32, 51)[0]
    return y

for name, fn in (("tiny add_", chain_tiny), ("add_layernorm 1632x256", chain_ln)):
    fn(); fn()
    e = timed(fn)
    s = torch.cuda.Stream()
    s.wait_stream(torch.cuda.current_stream())
    g = torch.cuda.CUDAGraph()
    with torch.cuda.stream(s):
        fn()
        torch.cuda.synchronize()
        with torch.cuda.graph(g, stream=s):
            fn()
    torch.cuda.synchronize()
    gr = timed(g.replay)
    print("%s: eager %.2f us/kernel, graph replay %.2f us/kernel" % (name, e / N, gr / N))
