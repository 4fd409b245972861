#!/usr/bin/env python3
"""The row-block data gradient (asr_dgrad_rows / asr_dgrad_rows_ln) against the tiled GEMM (+ the stand-alone LayerNorm backward) it
replaces, alone on the chip: HIP-event time of 50 back-to-back calls at the encoder's shapes."""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

from asr_amd import ops

DEV = "cuda:0"
B, L = 32, 1000
M = B * L


def timed(fn, n=50):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / n * 1e3


g = torch.Generator().manual_seed(0)
add = torch.randn(M, 256, generator=g).to(DEV)
s = torch.randn(M, 256, generator=g).to(DEV)
mean, rstd = s.mean(-1), 1.0 / torch.sqrt(s.var(-1, unbiased=False) + 1e-5)
gam = torch.ones(256, device=DEV)
lens = torch.full((B,), L, dtype=torch.int32, device=DEV)
dg, db, dbias = (torch.zeros(256, device=DEV) for _ in range(3))
for K in (256, 768, 3072):
    dy = (torch.randn(M, K, generator=g) * 0.1).bfloat16().to(DEV)
    w = (torch.randn(K, 256, generator=g) * 0.05).bfloat16().to(DEV)
    res = {}
    for rows in (True, False):
        ops.DGRAD_ROWS, ops.DGRAD_ROWS_MIN_K = rows, 64
        res[("f32+addend", rows)] = timed(lambda: ops.gemm_nn(dy, w, addend=add))
        res[("bf16", rows)] = timed(lambda: ops.gemm_nn(dy, w, out_dtype=torch.bfloat16))
    ops.DGRAD_ROWS = True
    res[("fold", True)] = timed(lambda: ops.gemm_nn_ln(dy, w, add, B, L, s, mean, rstd, gam, lens, dg, db, dbias=dbias))
    dx = ops.gemm_nn(dy, w, addend=add)
    res[("ln_bwd alone", False)] = timed(lambda: ops.add_layernorm_bwd(dx, s, mean, rstd, gam, lens, B, L, dg, db, want_bf16=True, dbias=dbias))
    print("K=%4d  f32+addend: rows %.1f us, tiled %.1f   bf16: rows %.1f, tiled %.1f   rows + LayerNorm backward folded %.1f (tiled + stand-alone %.1f + %.1f)" %
          (K, res[("f32+addend", True)], res[("f32+addend", False)], res[("bf16", True)], res[("bf16", False)], res[("fold", True)],
           res[("f32+addend", False)], res[("ln_bwd alone", False)]))
