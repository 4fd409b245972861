import os, sys, subprocess
ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
code = r'''
import sys, torch, numpy as np
sys.path.insert(0, %r)
import asr_amd
from asr_amd import ops
from asr_amd._lib import lib
B, h, Lq, Lk = 1, 1, 128, 128
torch.manual_seed(1)
q = (torch.randn(B, h, Lq, 64) * 0.6).cuda().bfloat16(); k = torch.randn(B, h, Lk, 64).cuda().bfloat16(); v = torch.randn(B, h, Lk, 64).cuda().bfloat16()
d = ops.Dropout(6554, 3, 4)
bits = ops.attention_dropmask(d, B, h, Lq, Lk, "cuda")
ctx, lse = ops.attention_fwd(q, k, v, None, False, need_lse=True, drop=d, drop_bits=bits)
dctx = torch.randn(B, Lq, h * 64).cuda().bfloat16()
dq = torch.zeros(B * Lq, h * 64, device="cuda", dtype=torch.bfloat16); dkv = torch.zeros(B * Lk, 2 * h * 64, device="cuda", dtype=torch.bfloat16)
ops.attention_bwd(q, k, v, ctx, dctx, lse, None, False, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:], drop=d, drop_bits=bits)
torch.save(dict(q=q.cpu(), k=k.cpu(), v=v.cpu(), ctx=ctx.cpu(), lse=lse.cpu(), dctx=dctx.cpu(), bits=bits.cpu(), dkv=dkv.cpu()), sys.argv[1])
''' % ROOT
import torch, numpy as np
outs = {}
for v4 in ("1", "0"):
    p = "/tmp/emu_%s.pt" % v4
    subprocess.run([sys.executable, "-c", code, p], check=True, env=dict(os.environ, ASR_AMD_ATTN_BWD_V4=v4), stderr=subprocess.DEVNULL)
    outs[v4] = torch.load(p)
z = outs["1"]
q, k, v, dO, O = (z[n].double()[0, 0] if z[n].dim() == 4 else z[n].double()[0] for n in ("q", "k", "v", "dctx", "ctx"))
lse = z["lse"].double()[0, 0]
Lq = Lk = 128
mk = z["bits"][:4 * 128].view(4, 128).numpy().view(np.uint32)       # Mk [kw][q]
keep = ((mk[:, :, None] >> np.arange(32, dtype=np.uint32)) & 1).transpose(1, 0, 2).reshape(128, 128).astype(np.float64)   # [q][key]
keep = torch.from_numpy(keep)
dsc = 65536.0 / (65536 - 6554)
S = q @ k.T
P = torch.exp2(S - lse[:, None])
delta = (dO * O).sum(-1)
dP = dO @ v.T
dS = P * (keep * dsc * dP - delta[:, None])
dS16 = dS.float().bfloat16().double()
dK = dS16.T @ q * 0.6931471805599453
Pk16 = (P * keep).float().bfloat16().double()
dV = Pk16.T @ dO * dsc
for name, ref, col in (("dk", dK, slice(0, 64)), ("dv", dV, slice(64, 128))):
    for v4 in ("1", "0"):
        got = outs[v4]["dkv"].double()[:, col]
        print(name, "asm" if v4 == "1" else "c++", "vs emulation: max %.4f rel-L2 %.5f" % (float((got - ref).abs().max()), float((got - ref).norm() / ref.norm())))
got = outs["1"]["dkv"].double()[:, 0:64]
def rep(name, dSx):
    dKx = dSx.float().bfloat16().double().T @ q * 0.6931471805599453
    print("  hypothesis %-34s rel-L2 vs asm %.5f" % (name, float((got - dKx).norm() / dKx.norm())))
rep("exact", dS)
rep("kept P in dS (dropped -> 0)", (P * keep) * (keep * dsc * dP - delta[:, None]))
rep("no dsc", P * (keep * dP - delta[:, None]))
rep("delta = 0", P * (keep * dsc * dP))
d_sw = delta.view(2, 2, 32)[:, [1, 0]].reshape(128)
rep("delta of the other half", P * (keep * dsc * dP - d_sw[:, None]))
d_t = delta.view(2, 64)[[1, 0]].reshape(128)
rep("delta of the other tile", P * (keep * dsc * dP - d_t[:, None]))
rep("mask not applied to dP", P * (dsc * dP - delta[:, None]))
keep_sw = keep.clone().view(2, 2, 32, 128)[:, [1, 0]].reshape(128, 128)
rep("dP masked by the other half's bits", P * (keep_sw * dsc * dP - delta[:, None]))
rep("dsc applied after delta", P * dsc * (keep * dP - delta[:, None]))
rep("delta scaled by dsc", P * (keep * dsc * dP - dsc * delta[:, None]))
b16 = lambda t: t.float().bfloat16().double()
rep("P rounded to bf16", b16(P) * (keep * dsc * dP - delta[:, None]))
rep("dP rounded to bf16", P * (keep * dsc * b16(dP) - delta[:, None]))
rep("delta rounded to bf16", P * (keep * dsc * dP - b16(delta)[:, None]))
rep("V scaled by dsc in bf16", P * (keep * (dO @ b16(v * dsc).T) - delta[:, None]))
rep("dS rounded twice", b16(dS))
rep("lse rounded to bf16", torch.exp2(S - b16(lse)[:, None]) * (keep * dsc * dP - delta[:, None]))
