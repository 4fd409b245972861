"""Cost of torch.cuda.Stream.wait_event on the main stream when the event (recorded on a side stream) has long completed."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
dev = torch.device("cuda:0")
x = torch.zeros(1 << 16, device=dev)
aux = torch.cuda.Stream(device=dev)
main = torch.cuda.current_stream()
def run(nwait, n=240):
    evs = []
    with torch.cuda.stream(aux):
        for _ in range(nwait):
            x2 = x + 1
            e = torch.cuda.Event(); e.record(aux); evs.append(e)
    aux.synchronize()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        if nwait and i % (n // nwait) == 0: main.wait_event(evs[i // (n // nwait)])
        x.add_(1.0)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3
for _ in range(2):
    for nw in (0, 12, 24, 48):
        print("waits=%d  %.1f us total for 240 tiny kernels" % (nw, run(nw)))

def run_rec(nrec, n=240):
    """event RECORDS on the main stream between kernels (first use of an event object creates the hipEvent: ~20 us each, pooled after)"""
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for i in range(n):
        x.add_(1.0)
        if nrec and i % (n // nrec) == 0:
            e = torch.cuda.Event(); e.record(main)
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) * 1e3
for _ in range(2):
    for nr in (0, 12, 48):
        print("records=%d  %.1f us total for 240 tiny kernels" % (nr, run_rec(nr)))
