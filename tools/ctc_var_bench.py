import os, sys, json
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import torch
from asr_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from bench_ops import timeit
DEV="cuda:0"
B, L, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
logits = torch.randn(B, L, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), L, dtype=torch.int32).to(DEV)
for nck in (1, 8, 16, 32, 64):
    t = timeit(lambda: ops.ctc_loss_fwd(logits, il, tg, n_chunks=nck))
    print(json.dumps(dict(var=os.environ.get("ASR_AMD_CTC_FUSED_VAR","0"), n_chunks=nck, ms=round(t,4))))
