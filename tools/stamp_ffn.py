"""Diagnostic: where the fused feed-forward forward's loop iteration spends its cycles (a -DFFN_STAMP build of csrc/ffn.hip: s_memtime
stamps around the pinned body and the end-of-iteration wait + barrier; read the SHARES, not the run time - stamps serialise)."""
import ctypes, os, subprocess, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
csrc = os.path.join(ROOT, "end-to-end_asr_pytorch_amd", "csrc")
so = "/tmp/libasr_stamp.so"
objs = [os.path.join(csrc, "build", f) for f in os.listdir(os.path.join(csrc, "build")) if f.endswith(".o") and not f.startswith("ffn")]
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-DFFN_STAMP", "-c", os.path.join(csrc, "ffn.hip"), "-o", "/tmp/ffn_stamp.o"])
subprocess.check_call(["/opt/rocm/bin/hipcc", "--offload-arch=gfx950", "-shared", "-fPIC", "-o", so, "/tmp/ffn_stamp.o"] + objs)
os.environ["ASR_AMD_LIB"] = so
import torch
import asr_amd
from asr_amd import ops
DEV = "cuda:0"
B, L, dff = 32, 1000, 2048
M = B * L
x32 = torch.randn(M, 256, device=DEV); x16 = x32.bfloat16()
w1 = (torch.randn(dff, 256, device=DEV) * 0.06).bfloat16(); w2 = (torch.randn(256, dff, device=DEV) * 0.03).bfloat16()
b1 = torch.randn(dff, device=DEV) * 0.1; b2 = torch.randn(256, device=DEV) * 0.1
gamma = torch.ones(256, device=DEV); beta = torch.zeros(256, device=DEV)
lens = torch.full((B,), L, device=DEV, dtype=torch.int32)
st = torch.zeros(1000 * 8, device=DEV, dtype=torch.int64)
L_ = ctypes.CDLL(so)
L_.asr_ffn_debug_stamps.argtypes = [ctypes.c_void_p]
asr_amd.lib().asr_ffn_debug_stamps = L_.asr_ffn_debug_stamps
for train in (False, True):
    for _ in range(3):
        asr_amd.lib().asr_ffn_debug_stamps(ctypes.c_void_p(st.data_ptr()))
        ops.ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=lens, train=train, drop_x=ops.Dropout(6554, 5, 9, None) if train else None)
    torch.cuda.synchronize()
    s = st.view(1000, 8).double().cpu()
    n = dff // 64 - 1
    print("train" if train else "eval", "per iteration (cycles, mean over waves): body %.0f  wait+barrier %.0f  [min/max body %.0f/%.0f wait %.0f/%.0f]" % (
        s[:, 0].mean() / n, s[:, 1].mean() / n, s[:, 0].min() / n, s[:, 0].max() / n, s[:, 1].min() / n, s[:, 1].max() / n))
    print("   phases (cycles, mean [min..max] over waves): prologue+chunk0 %.0f [%.0f..%.0f] | loop %.0f | last body %.0f | epilogue issue %.0f [%.0f..%.0f] | store drain %.0f [%.0f..%.0f]" % (
        s[:, 2].mean(), s[:, 2].min(), s[:, 2].max(), s[:, 3].mean(), s[:, 4].mean(), s[:, 5].mean(), s[:, 5].min(), s[:, 5].max(), s[:, 6].mean(), s[:, 6].min(), s[:, 6].max()))
    e = s[:, 7]
    print("   wave entry spread: %.0f cycles" % (e.max() - e.min()))
