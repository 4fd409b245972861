import os, sys, ctypes
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, asr_amd
from asr_amd._lib import lib, check
out = torch.zeros(256, dtype=torch.int16, device="cuda:0")
check(lib().asr_debug_probe_tr(ctypes.c_void_p(torch.cuda.current_stream().cuda_stream), ctypes.c_void_p(out.data_ptr()), 0, 0), "probe")
torch.cuda.synchronize()
v = out.cpu().numpy().astype(int) & 0xffff
for l in range(64):
    print("lane %2d (g%d i%2d):" % (l, l >> 4, l & 15), [(x >> 8, x & 255) for x in v[l * 4:l * 4 + 4]])
