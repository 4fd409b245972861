"""asr_proj_heads at S1's encoder size: the row-block kernel (ffn.hip) against the tiled GEMM, Q/K/V of a layer and the decoder's 6-layer cross K/V."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, L = 32, int(sys.argv[1]) if len(sys.argv) > 1 else 1000


def t(fn, n=30):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for n_proj in (3, 12):
    M, N = B * L, n_proj * 256
    x = torch.randn(M, 256, device=DEV).bfloat16(); w = (torch.randn(N, 256, device=DEV) * 0.06).bfloat16(); b = torch.randn(N, device=DEV) * 0.1
    res = {}
    for mode in ("0", "1"):
        os.environ["ASR_AMD_HEADS_ROWS"] = mode
        res[mode] = t(lambda: ops.proj_heads(x, w, b, n_proj, B, L, 4, 0.18))
    by = M * 256 * 2 + M * N * 2
    print("proj_heads M=%d N=%d: tiled %.1f us, rows %.1f us (%.0f GB/s of %d MB algorithmic, %.0f TF)" %
          (M, N, res["0"], res["1"], by / res["1"] / 1e3, by >> 20, 2.0 * M * N * 256 / res["1"] / 1e6))
