import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, L, U, V = 32, 1000, 50, 4234
g = torch.Generator().manual_seed(0)
logits = torch.randn(B, L, V, generator=g).to(DEV)
tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
il = torch.full((B,), L, dtype=torch.int32, device=DEV)
nck = int(sys.argv[1]) if len(sys.argv) > 1 else 4
for _ in range(6):
    ops.ctc_loss_fwd(logits, il, tg, n_chunks=nck)
torch.cuda.synchronize()
