#!/usr/bin/env python3
"""Fused feed-forward sub-layer (csrc/ffn.hip) against the separate launches it replaces: parity of every output and timing.
usage: python tools/bench_ffn.py [B L d_ff]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch

import asr_amd
from asr_amd import ops


def main():
    B, L, dff = (int(v) for v in sys.argv[1:4]) if len(sys.argv) >= 4 else (32, 1000, 2048)
    dev = torch.device("cuda:0")
    torch.manual_seed(0)
    M = B * L
    x32 = torch.randn(M, 256, device=dev)
    x16 = x32.bfloat16()
    w1 = (torch.randn(dff, 256, device=dev) * 0.06).bfloat16()
    w2 = (torch.randn(256, dff, device=dev) * 0.03).bfloat16()
    b1 = torch.randn(dff, device=dev) * 0.1
    b2 = torch.randn(256, device=dev) * 0.1
    gamma = 1 + 0.1 * torch.randn(256, device=dev)
    beta = 0.1 * torch.randn(256, device=dev)
    lens = torch.randint(L // 2, L + 1, (B,), device=dev, dtype=torch.int32)
    lens[0] = L
    asr_amd.manual_seed(7)
    from asr_amd.modules import dropout_site_keys, dropout_thr16
    k0, k1 = dropout_site_keys(7, "ffn.dropout", 1)
    drop = ops.Dropout(dropout_thr16(0.1), k0, k1, None)
    for dp in (None, drop):
        # separate launches
        bits_ref = ops.relu_bits_buffer(M, dff, dev)
        hid = ops.gemm_nt_ex(x16, w1, b1, out_dtype=torch.bfloat16, relu=True, relu_bits_out=bits_ref)
        o = ops.gemm_nt(hid, w2, b2)
        y32, y16, mean, rstd = ops.add_layernorm(o, x32, gamma, beta, B, L, row_len=lens, want_bf16=True, save_stats=True, drop_x=dp)
        f = ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=True, drop_x=dp)
        torch.cuda.synchronize()
        hid_f, bits_f, s_f, y32_f, y16_f, mean_f, rstd_f = f
        print("dropout" if dp else "no dropout")
        print("  hid   max|d| %.3e (|ref| max %.2f)" % ((hid_f.float() - hid.float()).abs().max().item(), hid.float().abs().max().item()))
        print("  s     max|d| %.3e" % (s_f - o).abs().max().item())
        print("  y32   max|d| %.3e" % (y32_f - y32).abs().max().item())
        print("  y16   max|d| %.3e" % (y16_f.float() - y16.float()).abs().max().item())
        print("  mean  max|d| %.3e   rstd max rel %.3e" % ((mean_f - mean).abs().max().item(), ((rstd_f - rstd) / rstd).abs().max().item()))
        # backward
        ds32 = torch.randn(M, 256, device=dev) * 0.01
        ds16 = ds32.bfloat16()
        d_hid = ops.gemm_nn(ds16, w2, out_dtype=torch.bfloat16, relu_bits=bits_ref)
        dx = ops.gemm_nn(d_hid, w1, addend=ds32)
        d_hid_f, dx_f = ops.ffn_bwd(ds16, ds32, w1, w2, bits_f)
        torch.cuda.synchronize()
        # the two paths mask with their own forward's bits (hid_f vs hid differ only in the last bf16 bit): compare where both agree
        both = (hid_f > 0) == (hid > 0)
        print("  mask agreement %.6f" % both.float().mean().item())
        print("  dhid  max|d| %.3e (|ref| max %.3f)" % (((d_hid_f.float() - d_hid.float()) * both).abs().max().item(), d_hid.float().abs().max().item()))
        print("  dx    max|d| %.3e (|ref| max %.3f)" % ((dx_f - dx).abs().max().item(), dx.abs().max().item()))
        # exact reference of the mask: dhid must vanish exactly where hid == 0
        print("  dhid nonzero where hid == 0: %d" % int(((d_hid_f != 0) & (hid_f == 0)).sum().item()))

    def timeit(fn, n=30):
        for _ in range(5):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(n):
            fn()
        e1.record()
        torch.cuda.synchronize()
        return e0.elapsed_time(e1) / n * 1e3

    bits_ref = ops.relu_bits_buffer(M, dff, dev)

    def sep_fwd():
        hid = ops.gemm_nt_ex(x16, w1, b1, out_dtype=torch.bfloat16, relu=True, relu_bits_out=bits_ref)
        o = ops.gemm_nt(hid, w2, b2)
        return ops.add_layernorm(o, x32, gamma, beta, B, L, row_len=lens, want_bf16=True, save_stats=True, drop_x=drop)

    def sep_bwd():
        d_hid = ops.gemm_nn(ds16, w2, out_dtype=torch.bfloat16, relu_bits=bits_ref)
        return ops.gemm_nn(d_hid, w1, addend=ds32)

    fl = 4.0 * M * 256 * dff
    t = timeit(sep_fwd)
    print("separate forward  %.1f us  (%.0f TF)" % (t, fl / t / 1e6))
    t = timeit(lambda: ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=True, drop_x=drop))
    print("fused forward (train)    %.1f us  (%.0f TF)" % (t, fl / t / 1e6))
    t = timeit(lambda: ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=False))
    print("fused forward (eval)     %.1f us  (%.0f TF)" % (t, fl / t / 1e6))
    t = timeit(sep_bwd)
    print("separate backward %.1f us  (%.0f TF)" % (t, fl / t / 1e6))
    t = timeit(lambda: ops.ffn_bwd(ds16, ds32, w1, w2, bits_f))
    print("fused backward    %.1f us  (%.0f TF)" % (t, fl / t / 1e6))


if __name__ == "__main__":
    main()
