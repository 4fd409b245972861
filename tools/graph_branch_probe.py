"""Does a replayed hipGraph run independent branches concurrently on this stack?  Two chains of low-occupancy kernels (each a
[256 x 256] x [256 x 4096] matmul series that fills ~1/8 of the chip) captured on two streams; eager two-stream, replay, serial."""
import time, torch
dev = torch.device("cuda:0")
a = [torch.randn(256, 256, device=dev, dtype=torch.bfloat16) for _ in range(2)]
b = [torch.randn(256, 4096, device=dev, dtype=torch.bfloat16) for _ in range(2)]
out = [torch.empty(256, 4096, device=dev, dtype=torch.bfloat16) for _ in range(2)]
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
N = 200
def chain(i):
    for _ in range(N):
        torch.matmul(a[i], b[i], out=out[i])
def both_streams():
    cur = torch.cuda.current_stream()
    s1.wait_stream(cur); s2.wait_stream(cur)
    with torch.cuda.stream(s1): chain(0)
    with torch.cuda.stream(s2): chain(1)
    cur.wait_stream(s1); cur.wait_stream(s2)
def serial():
    chain(0); chain(1)
def timeit(fn, n=5):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n * 1e3
print("eager serial      %.2f ms" % timeit(serial))
print("eager two streams %.2f ms" % timeit(both_streams))
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    both_streams()
print("replay two-branch graph %.2f ms" % timeit(g.replay))
g2 = torch.cuda.CUDAGraph()
with torch.cuda.graph(g2):
    serial()
print("replay serial graph     %.2f ms" % timeit(g2.replay))
