#!/usr/bin/env python3
"""What does the vendor library (torch.matmul -> hipBLASLt / rocBLAS) reach on the step's GEMM shapes?  A yardstick for the
hand-written kernels only; the product never calls it."""
import json
import torch
dev = "cuda:0"
M = 32000
def t(fn, n=30):
    for _ in range(5):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(n):
        fn()
    b.record(); torch.cuda.synchronize()
    return a.elapsed_time(b) / n
def bf(*s):
    return torch.randn(*s, device=dev).bfloat16()
cases = []
for (N, K) in ((2048, 256), (256, 2048), (768, 256), (256, 256), (4234, 256)):
    A, W = bf(M, K), bf(N, K)
    ms = t(lambda: torch.matmul(A, W.t()))
    cases.append(dict(op="nt", M=M, N=N, K=K, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1)))
for (N, K) in ((2048, 256), (256, 2048)):
    A, Bm = bf(M, K), bf(K, N)
    ms = t(lambda: torch.matmul(A, Bm))
    cases.append(dict(op="nn", M=M, N=N, K=K, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1)))
for (N, K) in ((2048, 256), (256, 2048), (256, 256)):
    dY, X = bf(M, N), bf(M, K)
    ms = t(lambda: torch.matmul(dY.t(), X))
    cases.append(dict(op="tn", M=M, N=N, K=K, ms=round(ms, 4), TF=round(2 * M * N * K / ms / 1e9, 1)))
for c in cases:
    print(json.dumps(c))
