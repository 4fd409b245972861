#!/usr/bin/env python3
"""Op-level microbenchmarks at the north-star shapes (one JSON line per op).  HIP-event timing on the launch stream.
Usage: python tools/bench_ops.py [ctc] [cif] [attn] [gemm] [ln]"""
import json
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import asr_amd  # noqa: E402
from asr_amd import ops  # noqa: E402

DEV = "cuda:0"


def timeit(fn, iters=20, warm=3):
    for _ in range(warm):
        fn()
    torch.cuda.synchronize()
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    for _ in range(iters):
        fn()
    b.record()
    torch.cuda.synchronize()
    return a.elapsed_time(b) / iters


def bench_ctc():
    B, L, U, V = 32, 1000, 50, 4234
    g = torch.Generator().manual_seed(0)
    logits = torch.randn(B, L, V, generator=g).to(DEV)
    tg = torch.randint(1, V - 1, (B, U), generator=g).to(DEV)
    il = torch.full((B,), L, dtype=torch.int32).to(DEV)
    gout = torch.ones(1, device=DEV)
    fwd = timeit(lambda: ops.ctc_loss_fwd(logits, il, tg))
    def fb():
        loss, nll, st = ops.ctc_loss_fwd(logits, il, tg)
        ops.ctc_loss_bwd(st, gout)
    both = timeit(fb)
    nb = B * L * V * 4
    for nck in (1, 4, 8, 12, 16, 24, 32):
        t = timeit(lambda: ops.ctc_loss_fwd(logits, il, tg, n_chunks=nck))
        print(json.dumps(dict(op="ctc_loss_fwd", n_chunks=nck, ms=round(t, 4), GBps=round(nb / t / 1e6, 1), frac_hbm_peak=round(nb / t / 1e6 / 8000, 4))))
    print(json.dumps(dict(op="ctc_loss", shape=[B, L, U, V], fwd_ms=round(fwd, 4), fwd_bwd_ms=round(both, 4),
                          fwd_GBps=round(nb / fwd / 1e6, 1), fwd_bwd_GBps=round(3 * nb / both / 1e6, 1),
                          fwd_frac_hbm_peak=round(nb / fwd / 1e6 / 8000, 4), fwd_bwd_frac_hbm_peak=round(3 * nb / both / 1e6 / 8000, 4))))


def bench_cif():
    B, L, H = 32, 1000, 256
    g = torch.Generator().manual_seed(0)
    a = torch.sigmoid(torch.randn(B, L, generator=g))
    U = torch.randint(20, 51, (B,), generator=g).float()
    a = (a * ((U + torch.rand(B, generator=g) - 0.5) / a.sum(-1))[:, None]).to(DEV)
    hid = torch.randn(B, L, H, generator=g).to(DEV)
    t_scan = timeit(lambda: ops.cif_scan(a, 0.95))
    cur, rem, fi, nf, nl = ops.cif_scan(a, 0.95)
    t_g = timeit(lambda: ops.cif_gather(hid, cur, rem, fi, nf, 50))
    nb = B * L * H * 4 + B * L * 4 + B * 50 * H * 4
    print(json.dumps(dict(op="cif", shape=[B, L, H], scan_us=round(t_scan * 1e3, 2), gather_us=round(t_g * 1e3, 2),
                          GBps=round(nb / (t_scan + t_g) / 1e6, 1), frac_hbm_peak=round(nb / (t_scan + t_g) / 1e6 / 8000, 4))))


def bench_attn():
    for (B, h, Lq, Lk, causal) in [(32, 4, 1000, 1000, False), (32, 4, 250, 250, False), (32, 4, 51, 1000, False), (32, 4, 51, 51, True)]:
        q = (torch.randn(B, h, Lq, 64, device=DEV) * 0.5).bfloat16()
        k = torch.randn(B, h, Lk, 64, device=DEV).bfloat16()
        v = torch.randn(B, h, Lk, 64, device=DEV).bfloat16()
        fl = 4.0 * B * h * 64 * Lq * Lk
        dd = ops.Dropout(6554, 1, 2)
        t = timeit(lambda: ops.attention_dropmask(dd, B, h, Lq, Lk, DEV))
        print(json.dumps(dict(op="attention_dropmask", shape=[B, h, Lq, Lk], us=round(t * 1e3, 2))))
        bits = ops.attention_dropmask(dd, B, h, Lq, Lk, DEV)
        for drop in (None, dd):
            t = timeit(lambda: ops.attention_fwd(q, k, v, None, causal, drop=drop, drop_bits=bits if drop else None))
            print(json.dumps(dict(op="attention_fwd", shape=[B, h, Lq, Lk], causal=causal, dropout=0.1 if drop else 0.0, us=round(t * 1e3, 2),
                                  TFLOPs=round(fl / t / 1e9, 1), frac_mfma_peak=round(fl / t / 1e9 / 2500, 4))))
        if Lq == 1000:
            ctx, lse = ops.attention_fwd(q, k, v, None, causal, need_lse=True)
            dctx = torch.randn_like(ctx)
            dq = torch.empty(B * Lq, h * 64, device=DEV, dtype=torch.bfloat16)
            dkv = torch.empty(B * Lk, 2 * h * 64, device=DEV, dtype=torch.bfloat16)
            for drop in (None, dd):
                t = timeit(lambda: ops.attention_bwd(q, k, v, ctx, dctx, lse, None, causal, 0.125, dq, dkv[:, :h * 64], dkv[:, h * 64:], drop=drop,
                                                     drop_bits=bits if drop else None))
                print(json.dumps(dict(op="attention_bwd(dq+dkv)", shape=[B, h, Lq, Lk], dropout=0.1 if drop else 0.0, us=round(t * 1e3, 2),
                                      TFLOPs=round(2.5 * fl / t / 1e9, 1), frac_mfma_peak=round(2.5 * fl / t / 1e9 / 2500, 4))))


def bench_gemm():
    for (M, N, K, relu, odt) in [(32000, 2048, 256, True, torch.bfloat16), (32000, 256, 2048, False, torch.float32),
                                 (32000, 256, 256, False, torch.float32), (32000, 4234, 256, False, torch.float32),
                                 (8192, 8192, 8192, False, torch.bfloat16)]:
        a = torch.randn(M, K, device=DEV).bfloat16()
        w = (torch.randn(N, K, device=DEV) / K ** 0.5).bfloat16()
        b = torch.randn(N, device=DEV)
        t = timeit(lambda: ops.gemm_nt(a, w, b, out_dtype=odt, relu=relu))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(op="gemm_nt", shape=[M, N, K], us=round(t * 1e3, 2), TFLOPs=round(fl / t / 1e9, 1),
                              frac_mfma_peak=round(fl / t / 1e9 / 2500, 4))))


def bench_bwdgemm():
    for (M, N, K) in [(32000, 256, 2048), (32000, 2048, 256), (32000, 768, 256), (32000, 256, 256), (1632, 256, 256), (1632, 2048, 256), (1632, 256, 2048)]:
        dy = torch.randn(M, N, device=DEV).bfloat16()
        x = torch.randn(M, K, device=DEV).bfloat16()
        out = torch.zeros(N, K, device=DEV)
        t = timeit(lambda: ops.gemm_tn(dy, x, out=out, accumulate=False))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(op="gemm_tn", shape=[M, N, K], us=round(t * 1e3, 2), TFLOPs=round(fl / t / 1e9, 1),
                              frac_mfma_peak=round(fl / t / 1e9 / 2500, 4))))
    # FFN hidden gradient: ReLU mask from the bf16 activation vs from sign bits
    M, N, K = 32000, 2048, 256
    dy = torch.randn(M, K, device=DEV).bfloat16()
    w = (torch.randn(K, N, device=DEV) / K ** 0.5).bfloat16()
    hid = torch.relu(torch.randn(M, N, device=DEV)).bfloat16()
    bits = torch.randint(0, 255, (M, N // 8), device=DEV, dtype=torch.uint8)
    for name, kw in (("plain", {}), ("relu_mask", dict(relu_mask=hid)), ("relu_bits", dict(relu_bits=bits))):
        t = timeit(lambda: ops.gemm_nn(dy, w, out_dtype=torch.bfloat16, **kw))
        print(json.dumps(dict(op="gemm_nn[32000,2048,256] bf16 out, " + name, us=round(t * 1e3, 2))))
    add = torch.randn(32000, 256, device=DEV)
    for K in (2048, 768, 256):
        dy2 = torch.randn(32000, K, device=DEV).bfloat16(); w2 = (torch.randn(K, 256, device=DEV) / K ** 0.5).bfloat16()
        t = timeit(lambda: ops.gemm_nn(dy2, w2, addend=add))
        print(json.dumps(dict(op="gemm_nn[32000,256,%d] f32 out + addend" % K, us=round(t * 1e3, 2))))
    for (M, N, K) in [(32000, 2048, 256), (32000, 256, 2048), (32000, 256, 768)]:
        dy = torch.randn(M, K, device=DEV).bfloat16()
        w = (torch.randn(K, N, device=DEV) / K ** 0.5).bfloat16()
        t = timeit(lambda: ops.gemm_nn(dy, w, out_dtype=torch.bfloat16))
        fl = 2.0 * M * N * K
        print(json.dumps(dict(op="gemm_nn", shape=[M, N, K], us=round(t * 1e3, 2), TFLOPs=round(fl / t / 1e9, 1),
                              frac_mfma_peak=round(fl / t / 1e9 / 2500, 4))))


def bench_ln():
    B, L, D = 32, 1000, 256
    x = torch.randn(B * L, D, device=DEV)
    r = torch.randn(B * L, D, device=DEV)
    g_, b_ = torch.ones(D, device=DEV), torch.zeros(D, device=DEV)
    t = timeit(lambda: ops.add_layernorm(x, r, g_, b_, B, L, want_bf16=True))
    nb = B * L * D * 14
    print(json.dumps(dict(op="add_layernorm", shape=[B * L, D], us=round(t * 1e3, 2), GBps=round(nb / t / 1e6, 1))))
    y32, _, mean, rstd = ops.add_layernorm(x.clone(), r, g_, b_, B, L, want_bf16=True, save_stats=True)
    dy = torch.randn(B * L, D, device=DEV)
    dg, db_, dbias = torch.zeros(D, device=DEV), torch.zeros(D, device=DEV), torch.zeros(D, device=DEV)
    t = timeit(lambda: ops.add_layernorm_bwd(dy, x, mean, rstd, g_, None, B, L, dg, db_, want_bf16=True, dbias=dbias))
    print(json.dumps(dict(op="add_layernorm_bwd", shape=[B * L, D], us=round(t * 1e3, 2), GBps=round(nb / t / 1e6, 1))))
    lens = torch.full((B,), L, dtype=torch.int32, device=DEV)
    dx = ops.Dropout(6554, 1, 2)
    xs = x.clone()
    t = timeit(lambda: ops.add_layernorm(xs, r, g_, b_, B, L, row_len=lens, want_bf16=True, save_stats=True, drop_x=dx))
    print(json.dumps(dict(op="add_layernorm(train: stats, mask, dropout)", us=round(t * 1e3, 2), GBps=round(B * L * D * 18 / t / 1e6, 1))))
    t = timeit(lambda: ops.add_layernorm_bwd(dy, x, mean, rstd, g_, lens, B, L, dg, db_, want_bf16=True, dbias=dbias, drop_x=dx))
    print(json.dumps(dict(op="add_layernorm_bwd(train: mask, dropout)", us=round(t * 1e3, 2), GBps=round(nb / t / 1e6, 1))))


if __name__ == "__main__":
    which = sys.argv[1:] or ["ctc", "cif", "attn", "gemm", "bwdgemm", "ln"]
    for w in which:
        globals()["bench_" + w]()
