#!/usr/bin/env python3
"""Does a CU-masked HIP stream (hipExtStreamCreateWithCUMask) confine kernels on this stack?  Times a large bf16 matmul on torch
ExternalStreams created with different masks (all 256 CUs, the low 128 bits, every other bit, the low 64 bits) and, with a masked
stream running a long kernel, the latency of a tiny kernel on the default stream."""
import ctypes
import json
import time

import torch

hip = ctypes.CDLL("libamdhip64.so")


def masked_stream(words):
    s = ctypes.c_void_p()
    arr = (ctypes.c_uint32 * len(words))(*words)
    rc = hip.hipExtStreamCreateWithCUMask(ctypes.byref(s), len(words), arr)
    assert rc == 0, rc
    return torch.cuda.ExternalStream(s.value), s


def timed(fn, stream, n=20):
    with torch.cuda.stream(stream):
        for _ in range(3):
            fn()
        stream.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(stream)
        for _ in range(n):
            fn()
        e1.record(stream)
        stream.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


def main():
    dev = torch.device("cuda", 0)
    a = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    b = torch.randn(8192, 8192, device=dev, dtype=torch.bfloat16)
    x = torch.randn(64, 64, device=dev)
    out = {}
    def low(n):
        return [(0xffffffff if n >= 32 * (i + 1) else ((1 << max(n - 32 * i, 0)) - 1)) for i in range(8)]

    def xcd_lt(k):      # bits whose index % 8 < k
        byte = (1 << k) - 1
        return [byte * 0x01010101] * 8

    masks = {"all256": [0xffffffff] * 8, "low128": low(128), "low64": low(64), "low192": low(192), "low224": low(224), "low240": low(240),
             "low248": low(248), "xcd_lt4": xcd_lt(4), "xcd_lt6": xcd_lt(6), "xcd_lt7": xcd_lt(7), "alt128": [0x55555555] * 8}
    keep = []
    for name, words in masks.items():
        st, raw = masked_stream(words)
        keep.append(raw)
        out[name + "_matmul_us"] = round(timed(lambda: torch.matmul(a, b), st), 1)
    # tiny kernel latency on the default stream while a masked stream is busy with the big matmul
    w1 = torch.randn(2048, 256, device=dev, dtype=torch.bfloat16)
    w2 = torch.randn(256, 2048, device=dev, dtype=torch.bfloat16)
    h0 = torch.randn(1632, 256, device=dev, dtype=torch.bfloat16)
    for name in ("all256", "low248", "low240", "low224", "low192", "low128", "xcd_lt7", "xcd_lt6", "xcd_lt4"):
        st, raw = masked_stream(masks[name])
        keep.append(raw)
        with torch.cuda.stream(st):
            for _ in range(40):
                torch.matmul(a, b)
        time.sleep(0.002)
        main_s = torch.cuda.current_stream()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(main_s)
        for _ in range(50):
            x = x * 1.0001
        e1.record(main_s)
        main_s.synchronize()
        out["tiny_chain50_us_while_" + name] = round(e0.elapsed_time(e1) * 1e3, 1)
        torch.cuda.synchronize()
        with torch.cuda.stream(st):
            for _ in range(40):
                torch.matmul(a, b)
        time.sleep(0.002)
        e0.record(main_s)
        h = h0
        for _ in range(25):
            h = torch.matmul(torch.matmul(h, w1.t()), w2.t()) * 0.01
        e1.record(main_s)
        main_s.synchronize()
        out["ffn_chain50_us_while_" + name] = round(e0.elapsed_time(e1) * 1e3, 1)
        torch.cuda.synchronize()
    print(json.dumps(out))


if __name__ == "__main__":
    main()
