"""The weight gradients of encoder layers (S1: 32 000 rows; per layer Q/K/V [768 x 256], output projection [256 x 256], feed-forward
[2048 x 256] and [256 x 2048] = 83.9 GFLOP, 80 output tiles) alone on the chip: one launch pair per weight (round 4) against ONE
batched launch of one / two layers at several workgroup budgets (asr_gemm_tn_ws_group_wgs; budget = tiles: every problem unsplit
over M, no slab, no reduce launch)."""
import os
import sys

ROOT = os.environ.get("GRAFT_REPO_ROOT", "/root/repo")
sys.path.insert(0, ROOT)
import torch

from asr_amd import ops

DEV = "cuda:0"
M = int(os.environ.get("M", "32000"))
g = torch.Generator().manual_seed(0)
SH = [(768, 256), (256, 256), (2048, 256), (256, 2048)]
FLOP_LAYER = sum(2.0 * M * n * k for n, k in SH)


def layer(seed):
    out = []
    for i, (n, k) in enumerate(SH):
        a = torch.randn(M, n, generator=g).to(DEV).bfloat16()
        b = torch.randn(M, k, generator=g).to(DEV).bfloat16()
        out.append((a, b, torch.zeros(n, k, device=DEV), True, torch.zeros(n, device=DEV) if i != 1 else None))
    return out


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


L1, L2, L3 = layer(0), layer(1), layer(2)
for wgs in (128, 256):
    us = timeit(lambda: [ops.gemm_tn(a, b, out=o, accumulate=acc, colsum=cs, max_wgs=wgs) for a, b, o, acc, cs in L1])
    print("single launches (max_wgs %3d), one layer:            %7.1f us  = %6.1f TF" % (wgs, us, FLOP_LAYER / us / 1e6))
for name, probs in (("one layer ", L1), ("two layers", L1 + L2), ("three     ", L1 + L2 + L3)):
    tiles = sum(ops.tn_tiles(p[0], p[1]) for p in probs)
    nl = len(probs) // 4
    for budget in sorted({tiles, 128, 192, 256, 2 * tiles} - {0}):
        if budget < 64:
            continue
        us = timeit(lambda: ops.gemm_tn_group(probs, group_wgs=budget))
        print("batched %s (%3d tiles), budget %3d workgroups:  %7.1f us  = %6.1f us per layer = %6.1f TF" % (
            name, tiles, budget, us, us / nl, nl * FLOP_LAYER / us / 1e6))
