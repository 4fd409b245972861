"""Ordering probe for the collective callback of the graph executor: after a hipGraphLaunch on the null stream (a memcpy node of a plan),
kernels torch queues through ExternalStream(0) overtake kernels hipLaunchKernel put on the null stream before them; the same work on
torch.cuda.default_stream() ("plain"), on a non-null stream ("side", "ns") or through the HIP API ("hip") stays ordered.
Mode = producer/callback; prints the number of mis-ordered replays per mode."""
import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
import torch
import asr_amd
from asr_amd import ops
dev = "cuda:0"
N = 64 << 20
hip = ctypes.CDLL("libamdhip64.so")
hip.hipMemcpyAsync.argtypes = [ctypes.c_void_p, ctypes.c_void_p, ctypes.c_size_t, ctypes.c_int, ctypes.c_void_p]
print("torch current stream handle", torch.cuda.current_stream().cuda_stream, ops._stream())
for mode in ("kernel/torch", "kernel/plain", "copy/plain", "kernel/plain", "kernel/torch", "kernel/hip", "kernel/plain"):
    prod, cbk = mode.split("/")
    src = torch.zeros(N, device=dev)
    dst = torch.zeros(N, device=dev)
    snap = torch.zeros(512, device=dev)
    other = torch.zeros(N, device=dev)
    side = torch.cuda.Stream()
    g = torch.cuda.CUDAGraph(keep_graph=True)
    run_stream = torch.cuda.Stream() if prod == "ns" else None
    with torch.cuda.graph(g):
        if prod == "copy":
            dst.copy_(src)
        else:
            torch.add(src, 0.0, out=dst)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.collective_mark(dst[-512:], tag=1)
        if prod == "side":            # a longer main chain beside the marker: the marker becomes a side branch
            for _ in range(3):
                other.add_(1.0)
        torch.cuda.current_stream().wait_stream(side)
        out = dst[-512:] + 0
    gx = ops.GraphExec.from_torch_graph(g)
    print(mode, gx.info)

    def fn(ctx, ptr, count, tag, stream):
        if cbk == "plain":
            torch.add(dst[-512:], 0.0, out=snap)
        elif cbk.startswith("torch"):
            with torch.cuda.stream(torch.cuda.ExternalStream(stream or 0)):
                torch.add(dst[-512:], 0.0, out=snap)
                if cbk == "torchmul":
                    dst[-512:].mul_(2.0)
        else:
            hip.hipMemcpyAsync(snap.data_ptr(), dst[-512:].data_ptr(), 2048, 3, stream)
        return 0
    cb = ops._COLLECTIVE_CB(fn); cb.state = {"error": None}
    gx.set_collective(fn=cb)
    bad = 0
    for i in range(1, 30):
        src.fill_(float(i))
        if run_stream is not None:
            run_stream.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(run_stream):
                gx.launch()
        else:
            gx.launch()
        torch.cuda.synchronize()
        k = 2 if cbk == "torchmul" else 1
        ok = float(snap[0]) == i and float(out[0]) == k * i
        bad += not ok
        if not ok and bad < 4:
            print("   replay", i, "snap", float(snap[0]), "out", float(out[0]), "dst", float(dst[-1]))
    print(mode, "bad replays:", bad)
