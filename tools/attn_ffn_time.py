import sys, torch
sys.path.insert(0, "/root/repo")
import asr_amd
from asr_amd import ops
dev = "cuda:0"
B, L, dff = 32, 1000, 2048
M = B * L
g = torch.Generator().manual_seed(1)
r = lambda *s: torch.randn(*s, generator=g)
ctx = r(M, 256).bfloat16().to(dev); x32 = r(M, 256).to(dev)
wo = (r(256, 256) * 0.06).bfloat16().to(dev); bo = (r(256) * 0.2).to(dev); g0 = (torch.rand(256) + 0.5).to(dev); be0 = (r(256) * 0.3).to(dev)
w1 = (r(dff, 256) * 0.06).bfloat16().to(dev); w2 = (r(256, dff) * 0.05).bfloat16().to(dev); b1 = (r(dff) * 0.2).to(dev); b2 = (r(256) * 0.2).to(dev)
gam = (torch.rand(256) + 0.5).to(dev); bet = (r(256) * 0.3).to(dev)
rl = torch.full((B,), L, dtype=torch.int32, device=dev)
d0, d1 = ops.Dropout(6554, 11, 3), ops.Dropout(6554, 5, 9)
def two():
    s0, y32, y16, m0, r0 = ops.proj_ln(ctx, wo, bo, x32, g0, be0, B, L, row_len=rl, save_stats=True, drop_x=d0, save_s=False)
    return ops.ffn_fwd(y16, y32, w1, b1, w2, b2, gam, bet, B, L, row_len=rl, train=True, drop_x=d1, save_s=False)
def one():
    return ops.attn_ffn_fwd(ctx, wo, bo, x32, g0, be0, w1, b1, w2, b2, gam, bet, B, L, row_len=rl, train=True, drop0=d0, drop_x=d1, save_s=False)
for name, fn in (("two launches", two), ("one launch", one), ("two launches", two), ("one launch", one)):
    for _ in range(5): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(30): fn()
    e1.record(); torch.cuda.synchronize()
    print("%s: %.1f us" % (name, e0.elapsed_time(e1) / 30 * 1e3))
