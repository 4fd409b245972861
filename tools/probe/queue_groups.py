import sys, ctypes, torch
sys.path.insert(0, "/root/repo")
import asr_amd
from asr_amd import ops
from asr_amd._lib import lib
dev = torch.device("cuda:0")
main = torch.cuda.Stream(dev)
streams = [torch.cuda.Stream(dev) for _ in range(10)]
def share(a, b):
    sh = ctypes.c_int()
    rc = lib().asr_streams_share_queue(ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream), ctypes.byref(sh))
    assert rc == 0
    return sh.value
with torch.cuda.stream(main):
    torch.zeros(1, device=dev)
groups = []
for i, s in enumerate([main] + streams):
    for g in groups:
        if share(g[0][1], s):
            g.append((i, s)); break
    else:
        groups.append([(i, s)])
print("queue groups (0 = main):", [[i for i, _ in g] for g in groups])
cur = torch.cuda.current_stream()
print("default stream shares with main:", share(cur, main), [share(cur, s) for s in streams[:4]])
