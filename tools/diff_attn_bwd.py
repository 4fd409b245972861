"""dK / dV of the generated kernel against attention_bwd.hip's C++ kernel on the same inputs (two processes: the switch is read once)."""
import os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
code = r'''
import sys, torch
sys.path.insert(0, %r)
import asr_amd
from asr_amd import ops
B, h, Lq, Lk, drop = (int(x) for x in sys.argv[2:7])
torch.manual_seed(1)
q = (torch.randn(B, h, Lq, 64) * 0.6).cuda().bfloat16(); k = torch.randn(B, h, Lk, 64).cuda().bfloat16(); v = torch.randn(B, h, Lk, 64).cuda().bfloat16()
kl = torch.tensor([Lk] + [max(1, Lk - 37 * (i + 1)) for i in range(B - 1)], dtype=torch.int32).cuda()
d = ops.Dropout(6554, 3, 4) if drop else None
ctx, lse = ops.attention_fwd(q, k, v, kl, False, need_lse=True, drop=d)
dctx = torch.randn(B, Lq, h * 64).cuda().bfloat16()
dq = torch.zeros(B * Lq, h * 64, device="cuda", dtype=torch.bfloat16); dkv = torch.zeros(B * Lk, 2 * h * 64, device="cuda", dtype=torch.bfloat16)
ops.attention_bwd(q, k, v, ctx, dctx, lse, kl, False, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:], drop=d)
torch.save((dq.cpu(), dkv.cpu()), sys.argv[1])
''' % ROOT
args = sys.argv[1:6] if len(sys.argv) > 5 else ["2", "4", "200", "200", "1"]
outs = []
for v4 in ("1", "0"):
    path = "/tmp/bwd_v4_%s.pt" % v4
    subprocess.run([sys.executable, "-c", code, path] + args, check=True, env=dict(os.environ, ASR_AMD_ATTN_BWD_V4=v4), stderr=subprocess.DEVNULL)
    import torch
    outs.append(torch.load(path))
h = int(args[1])
for name, a, b in (("dk", outs[0][1][:, :h * 64], outs[1][1][:, :h * 64]), ("dv", outs[0][1][:, h * 64:], outs[1][1][:, h * 64:])):
    e = (a.float() - b.float()).abs()
    rows = (e > 0.02).any(1).nonzero().flatten()
    print(name, "max |asm - c++| %.4f" % float(e.max()), " elements differing at all: %d of %d" % (int((e > 0).sum()), e.numel()), " rows with > 0.02:", rows[:20].tolist(),
          " cols:", (e > 0.02).any(0).nonzero().flatten()[:16].tolist())
