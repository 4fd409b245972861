"""asr_ffn_fwd (ffn2.hip: the generated two-waves-per-SIMD loop) against a torch-fp32 evaluation, error by output, then its time at the
S1 shape.  usage: python tools/check_ffn2.py [--time-only]"""
import os
import sys
import numpy as np
import torch
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests"))
import asr_amd
from asr_amd import ops
from test_gpu_ffn_fused import _case, _reference, THR
from oracle import asr_oracle as O

DEV = "cuda:0"
N = lambda t: t.detach().float().cpu().numpy()
d = lambda t: t.to(DEV).contiguous()


def check(B, L, dff, drop):
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=B * 1000 + L)
    M = B * L
    mask = torch.from_numpy(O.dropout_mask((B, L, 256), THR, 5, 9)).view(M, 256) if drop else None
    hid_ref, s_ref, y_ref = _reference(x32, w1, b1, w2, b2, gam, bet, lens, B, L, mask)
    for train in (True, False):
        hid, bits, s, y32, y16, mean, rstd = ops.ffn_fwd(d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L,
                                                         row_len=d(lens).int(), train=train, drop_x=ops.Dropout(THR, 5, 9) if drop else None)
        torch.cuda.synchronize()
        e = {"y32": float(np.abs(N(y32) - y_ref.detach().numpy()).max())}
        if train:
            e["hid"] = float(np.abs(N(hid) - hid_ref.detach().numpy()).max())
            e["s"] = float(np.abs(N(s) - s_ref.detach().numpy()).max())
            eh = np.abs(N(hid) - hid_ref.detach().numpy())
            bad = np.argwhere(eh > 5e-2)
            if len(bad):
                e["hid_bad"] = "%d of %d; first rows/cols %s" % (len(bad), eh.size, bad[:6].tolist())
        print("B %d L %d dff %d drop %d train %d: %s" % (B, L, dff, drop, train, e), flush=True)


def timeit(train, drop, n=30, dff=2048):
    B, L = 32, 1000
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=1)
    args = (d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L)
    kw = dict(row_len=d(lens).int(), train=train, drop_x=ops.Dropout(THR, 5, 9) if drop else None)
    for _ in range(5):
        ops.ffn_fwd(*args, **kw)
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        torch.cuda.synchronize()
        e0.record()
        for _ in range(n):
            ops.ffn_fwd(*args, **kw)
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / n * 1e3)
    print("ffn_fwd [32000 x 256 x %d] train %d drop %d: %.1f us = %.0f TF" % (dff, train, drop, best, 2 * 2.0 * 32000 * 256 * dff / best / 1e6), flush=True)


def stamps():
    """--stamps (a tools/gen_ffn_fwd.py --abl 64 build): cycles and 100-MHz ticks the loop took, per wave"""
    B, L, dff = 32, 1000, 2048
    x32, w1, b1, w2, b2, gam, bet, lens = _case(B, L, dff, seed=1)
    args = (d(x32.bfloat16()), d(x32), d(w1), d(b1), d(w2), d(b2), d(gam), d(bet), B, L)
    for _ in range(20):
        hid = ops.ffn_fwd(*args, row_len=d(lens).int(), train=True, drop_x=ops.Dropout(THR, 5, 9))[0]
    torch.cuda.synchronize()
    w = hid.view(torch.int32).view(B * L, dff // 2)[:, :4].cpu().numpy().astype(np.int64) & 0xffffffff
    w = w[::8]                      # one stamp per group of 8 rows (lanes with piece 0)
    cyc = (w[:, 2] - w[:, 0]) & 0xffffffff
    tick = (w[:, 3] - w[:, 1]) & 0xffffffff
    ok = (tick > 0) & (tick < 100000)
    cyc, tick = cyc[ok], tick[ok]
    print("loop: cycles median %d (min %d max %d), 100-MHz ticks median %d -> %.2f us, clock %.2f GHz; per chunk %.0f cycles" %
          (np.median(cyc), cyc.min(), cyc.max(), np.median(tick), np.median(tick) / 100.0, np.median(cyc) / np.median(tick) / 10.0, np.median(cyc) / (dff // 64)), flush=True)


if __name__ == "__main__":
    if "--stamps" in sys.argv:
        stamps()
        sys.exit(0)
    if "--time-only" not in sys.argv:
        for c in [(1, 5, 64, False), (4, 37, 128, False), (2, 128, 256, True), (5, 129, 512, True), (3, 100, 2048, True)]:
            check(*c)
    for dff in (64, 1024, 2048):      # (64: prologue + one chunk + epilogue; the difference to 2048 is 31 chunks of the loop)
        timeit(False, False, dff=dff)
        timeit(True, True, dff=dff)
