"""Thin tensor-level wrappers over the C-ABI (include/asr_hip.h).

Every function here takes torch CUDA tensors, passes raw device pointers + torch's current HIP stream to
libasr_hip.so and returns torch tensors that own the outputs.  torch is plumbing only (allocation, streams);
no arithmetic on the product path is done by torch ops.  There is no CPU fallback: CPU tensors raise.
"""
import atexit
import ctypes
import weakref
import os
import threading

import torch

from ._lib import NO_DROP, Dropout, check, lib

F32, BF16, F16 = 0, 1, 2
GEMM_RELU = 1


def dtype_code(t):
    if t.dtype == torch.float32:
        return F32
    if t.dtype == torch.bfloat16:
        return BF16
    if t.dtype == torch.float16:      # (the CTC branch's logits image only)
        return F16
    raise TypeError("asr_amd: unsupported dtype %s" % t.dtype)


def torch_dtype(code):
    return torch.float32 if code == F32 else torch.bfloat16


def _p(t):
    return None if t is None else ctypes.c_void_p(t.data_ptr())


_raw_stream = getattr(torch._C, "_cuda_getCurrentRawStream", None)
_raw_device = getattr(torch._C, "_cuda_getDevice", None)


def _stream():
    """torch's current HIP stream of the current device as a raw handle.  Through the two C entry points torch itself uses: one
    `torch.cuda.current_stream()` builds a Stream object (device-index resolution, an is_available() check, os.environ reads -
    ~6 us), and an S1 training step asks ~590 times: 3-4 ms of its ~11 ms of host queueing work."""
    if _raw_stream is not None and _raw_device is not None:
        return ctypes.c_void_p(_raw_stream(_raw_device()))
    return ctypes.c_void_p(torch.cuda.current_stream().cuda_stream)


def _req_cuda(*ts):
    for t in ts:
        if t is not None and not t.is_cuda:
            raise RuntimeError("asr_amd: tensor is on %s - the MI355X path has no CPU fallback" % t.device)


# ---- optional live per-kernel timing (bench.py): HIP events on the launch stream around each C-ABI call ----------
_PROF = None


_PROF_SEQ = None


def profile_start(sequence=False):
    """sequence=True also keeps the bracketed calls in issue order (profile_sequence: a time line of one step, every stream)."""
    global _PROF, _PROF_SEQ
    _PROF = {}
    _PROF_SEQ = [] if sequence else None


def profile_mark():
    """a timing event on the current stream: the origin of profile_sequence's time axis"""
    return _timer()


def profile_sequence(origin):
    """-> [(name, start_ms, duration_ms, stream)] in issue order; call before profile_stop, after a device synchronize."""
    return [(n, _elapsed_ms(origin, a), _elapsed_ms(a, b), st) for n, a, b, st in (_PROF_SEQ or [])]


LIGHT_TIMERS = True      # timing events without the system-scope fence (False: torch timing events, ~2 us more per bracketed op)
_TIMER_POOL = []


def _timer():
    """a timing event without the system-scope fence (asr_hip.h: asr_timer_create), recorded on the current stream"""
    if _TIMER_POOL:
        h = _TIMER_POOL.pop()
    else:
        h = ctypes.c_void_p()
        check(lib().asr_timer_create(ctypes.byref(h)), "asr_timer_create")
    check(lib().asr_timer_record(h, _stream()), "asr_timer_record")
    return h


def _elapsed_ms(a, b):
    if isinstance(a, ctypes.c_void_p):
        ms = ctypes.c_float()
        check(lib().asr_timer_elapsed_ms(a, b, ctypes.byref(ms)), "asr_timer_elapsed_ms")
        return float(ms.value)
    return a.elapsed_time(b)


def profile_stop():
    """-> {name: dict(calls, ms, work)} ; `work` = algorithmic FLOPs (MFMA kernels) or bytes (HBM kernels) summed."""
    global _PROF
    prof, _PROF = _PROF, None
    torch.cuda.synchronize()
    out = {}
    for name, recs in (prof or {}).items():
        out[name] = dict(calls=len(recs), ms=sum(_elapsed_ms(a, b) for a, b, _ in recs), work=sum(w for _, _, w in recs))
        for a, b, _ in recs:
            if isinstance(a, ctypes.c_void_p):
                _TIMER_POOL.extend((a, b))
    return out


class _timed:
    __slots__ = ("name", "work", "a")

    def __init__(self, name, work):
        self.name, self.work = name, work

    def __enter__(self):
        if _PROF is not None:
            if LIGHT_TIMERS:
                self.a = _timer()
            else:
                self.a = torch.cuda.Event(enable_timing=True)
                self.a.record()

    def __exit__(self, *exc):
        if _PROF is not None:
            if LIGHT_TIMERS:
                b = _timer()
            else:
                b = torch.cuda.Event(enable_timing=True)
                b.record()
            _PROF.setdefault(self.name, []).append((self.a, b, self.work))
            if _PROF_SEQ is not None:
                _PROF_SEQ.append((self.name, self.a, b, _stream().value or 0))
        return False


def as_i32(t, device=None):
    t = t.to(device=device if device is not None else t.device, dtype=torch.int32)
    return t.contiguous()


# ------------------------------------------------------------------------------------------------------------
# ---- zero arena -------------------------------------------------------------------------------------------------
# The decoder-sized GEMMs are split over K and accumulate with atomics into a zeroed C (gemm.hip: pick_ksplit); 19 of them per
# training step would each launch a zeroing kernel on the latency-bound decoder chain (a dependent launch costs ~5 us whatever it
# does).  The trainer zeroes ONE arena at the start of the step instead; GEMM outputs of split-K shape are cut from it and the
# library is told (ASR_GEMM_C_IS_ZERO).  Every slice is handed out once per reset, so it is still zero when the GEMM runs.
_ARENA = {"buf": None, "off": 0, "live": False, "dirty": 0}
SPLITK_TILES = 64     # gemm.hip pick_ksplit: output tiles up to which K is split
GEMM_C_IS_ZERO = 4


def arena_reset(device, nbytes=64 << 20):
    """Zero the arena (one fill launch) and make its slices available until the next reset.  Call once per step, on the stream
    the GEMMs will run on."""
    device = torch.device(device)
    if _ARENA["buf"] is None or _ARENA["buf"].device != device or _ARENA["buf"].numel() * 4 < nbytes:
        _ARENA["buf"] = torch.zeros(nbytes // 4, device=device, dtype=torch.float32)
        _ARENA["dirty"] = 0
    if _ARENA["dirty"]:
        _ARENA["buf"][:_ARENA["dirty"]].zero_()      # only what was handed out since the last reset
    _ARENA["off"], _ARENA["dirty"], _ARENA["live"] = 0, 0, True


def arena_release():
    _ARENA["live"] = False     # (slices already handed out stay valid: they are views of the arena tensor)


def _arena_take(M, N, K, device):
    """-> zeroed f32 [M, N] for a GEMM output that the library will split over K, or None."""
    if not _ARENA["live"] or _ARENA["buf"].device != device:
        return None
    tiles = ((M + 127) // 128) * ((N + 127) // 128)
    if tiles > SPLITK_TILES or K < 512 or N % 128 != 0:       # mirror of gemm.hip pick_ksplit (a mismatch only costs the zeroing launch)
        return None
    n = M * N
    off = (_ARENA["off"] + 63) // 64 * 64
    if off + n > _ARENA["buf"].numel():
        return None
    _ARENA["off"] = _ARENA["dirty"] = off + n
    return _ARENA["buf"][off:off + n].view(M, N)


def gemm_nt_raw(a, M, K, lda, w, bias=None, out_dtype=torch.float32, relu=False, out=None, ldc=None):
    """C[M,N] = act(A . W^T + bias) where A is addressed as rows of K elements with stride lda inside tensor `a`."""
    _req_cuda(a, w, bias)
    N = w.shape[0]
    assert w.dim() == 2 and w.shape[1] == K and w.is_contiguous()
    flags = GEMM_RELU if relu else 0
    if out is None:
        out = _arena_take(M, N, K, a.device) if (out_dtype == torch.float32 and not relu and a.dtype == torch.bfloat16) else None
        if out is not None:
            flags |= GEMM_C_IS_ZERO
        else:
            out = torch.empty((M, N), device=a.device, dtype=out_dtype)
        ldc = N
    with _timed("gemm_nt[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
        check(lib().asr_gemm_nt(_stream(), _p(a), dtype_code(a), lda, _p(w), dtype_code(w), K, _p(bias), _p(out), dtype_code(out),
                                ldc if ldc is not None else N, M, N, K, flags), "asr_gemm_nt")
    return out


def gemm_nt(a2d, w, bias=None, out_dtype=torch.float32, relu=False):
    assert a2d.dim() == 2 and a2d.is_contiguous()
    return gemm_nt_raw(a2d, a2d.shape[0], a2d.shape[1], a2d.shape[1], w, bias, out_dtype, relu)


def proj_heads(x2d, w, bias, n_proj, B, L, h, scale_first=1.0):
    """-> tensor [n_proj, B, h, L, 64] in w.dtype (head-major), projection 0 scaled by scale_first."""
    _req_cuda(x2d, w, bias)
    assert x2d.is_contiguous() and w.is_contiguous() and x2d.shape[0] == B * L and w.shape[0] == n_proj * h * 64
    K = x2d.shape[1]
    out = torch.empty((n_proj, B, h, L, 64), device=x2d.device, dtype=w.dtype)
    with _timed("proj_heads[%dx%dx%d]" % (B * L, n_proj * h * 64, K), 2.0 * B * L * n_proj * h * 64 * K):
        check(lib().asr_proj_heads(_stream(), _p(x2d), dtype_code(x2d), K, _p(w), dtype_code(w), K, _p(bias), _p(out),
                                   B * h * L * 64, n_proj, B, L, h, K, float(scale_first)), "asr_proj_heads")
    return out


def _d(drop):
    return NO_DROP if drop is None else drop


def dropout_apply(x, drop, N0, N1, N2, out=None):
    """y = dropout(x) over f32 [N0,N1,N2] with the counter-based mask of `drop` (in place by default)."""
    _req_cuda(x)
    assert x.dtype == torch.float32 and x.is_contiguous() and x.numel() == N0 * N1 * N2
    y = x if out is None else out
    check(lib().asr_dropout_apply(_stream(), _p(x), _p(y), N0, N1, N2, drop), "asr_dropout_apply")
    return y


def attention_dropmask(drop, B, h, Lq, Lk, device):
    """Keep-bit images of one attention call's dropout (asr_hip.h: asr_attention_dropmask): hashed once, read by the forward and
    both backward kernels.  -> uint32 tensor, or None when `drop` is off."""
    if drop is None or drop.thr16 == 0:
        return None
    bits = torch.empty((int(lib().asr_attention_dropmask_words(B, h, Lq, Lk)),), device=device, dtype=torch.int32)
    with _timed("attention_dropmask[B%d h%d %dx%d]" % (B, h, Lq, Lk), 0.0):
        check(lib().asr_attention_dropmask(_stream(), drop, B, h, Lq, Lk, _p(bits)), "asr_attention_dropmask")
    return bits


def attention_dropmask_multi(drops, B, h, Lq, Lk, device):
    """attention_dropmask for up to 8 sites of one shape in one launch -> list of uint32 tensors (views of one allocation)."""
    n = len(drops)
    assert 0 < n <= 8 and all(d is not None and d.thr16 > 0 for d in drops)
    words = int(lib().asr_attention_dropmask_words(B, h, Lq, Lk))
    buf = torch.empty((n, words), device=device, dtype=torch.int32)
    darr = (Dropout * n)(*drops)
    parr = (ctypes.c_void_p * n)(*[buf[i].data_ptr() for i in range(n)])
    with _timed("attention_dropmask[%dx B%d h%d %dx%d]" % (n, B, h, Lq, Lk), 0.0):
        check(lib().asr_attention_dropmask_multi(_stream(), n, ctypes.cast(darr, ctypes.c_void_p), ctypes.cast(parr, ctypes.c_void_p),
                                                 B, h, Lq, Lk), "asr_attention_dropmask_multi")
    return [buf[i] for i in range(n)]


def attention_fwd(q, k, v, k_len=None, causal=False, need_lse=False, drop=None, drop_bits=None):
    """q [B,h,Lq,64] (pre-scaled), k/v [B,h,Lk,64] -> ctx [B,Lq,h*64] (same dtype), lse [B,h,Lq] or None.
    drop_bits: attention_dropmask(...) of this call (made here when omitted; pass it to share it with attention_bwd)."""
    _req_cuda(q, k, v, k_len)
    B, h, Lq, dk = q.shape
    Lk = k.shape[2]
    if drop_bits is None:
        drop_bits = attention_dropmask(drop, B, h, Lq, Lk, q.device)
    assert dk == 64 and q.is_contiguous() and k.is_contiguous() and v.is_contiguous()
    ctx = torch.empty((B, Lq, h * 64), device=q.device, dtype=q.dtype)
    lse = torch.empty((B, h, Lq), device=q.device, dtype=torch.float32) if need_lse else None
    with _timed("attention_fwd[B%d h%d %dx%d]" % (B, h, Lq, Lk), 4.0 * B * h * 64 * Lq * Lk):
        check(lib().asr_attention_fwd(_stream(), _p(q), _p(k), _p(v), dtype_code(q), _p(ctx), _p(lse), B, h, Lq, Lk, _p(k_len),
                                      1 if causal else 0, _d(drop), _p(drop_bits)), "asr_attention_fwd")
    return ctx, lse


SMALL_FUSED_MAX_ROWS = 4096     # 0 switches the fused small-M block off


def gemm_add_layernorm_small_ok(a2d, w, D, B, L):
    """Shapes asr_gemm_add_layernorm_small takes (and is meant for)."""
    # K <= 512: a workgroup walks K alone (one round trip to L2 per 256 of K, ~0.6 us each: at K = 2048 the split-K GEMM + LayerNorm
    # pair is faster - 42 vs 52 us)
    return (0 < B * L <= SMALL_FUSED_MAX_ROWS and D == 256 and w.shape[0] == 256 and a2d.dtype == torch.bfloat16 and
            w.dtype == torch.bfloat16 and a2d.shape[1] % 32 == 0 and a2d.shape[1] <= 512 and a2d.is_contiguous() and w.is_contiguous())


def gemm_add_layernorm_ok(a2d, w, D, B, L):
    """Shapes the fused projection + LayerNorm launch takes (decoder-sized rows; the encoder-sized variant of round 2 was slower than
    the GEMM + LayerNorm pair and is gone - the encoder's feed-forward sub-layer is ffn_fwd)."""
    return gemm_add_layernorm_small_ok(a2d, w, D, B, L)


def gemm_add_layernorm_small(a2d, w, bias, residual, gamma, beta, B, L, row_len=None, want_bf16=True, eps=1e-5, save_stats=False,
                             drop_x=None):
    """LayerNorm(dropout_x(A . W^T + bias) + residual) for decoder-sized rows in one launch (asr_hip.h).
    -> (s_sum [M,256] pre-norm sum, y32, y16 or None, mean, rstd) - the tensors the unfused gemm_nt + add_layernorm pair leaves."""
    _req_cuda(a2d, w, bias, residual, gamma, beta, row_len)
    M, K = a2d.shape
    assert M == B * L
    dev = a2d.device
    s_sum = torch.empty((M, 256), device=dev, dtype=torch.float32)
    y32 = torch.empty((M, 256), device=dev, dtype=torch.float32)
    y16 = torch.empty((M, 256), device=dev, dtype=torch.bfloat16) if want_bf16 else None
    mean = torch.empty(M, device=dev, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=dev, dtype=torch.float32) if save_stats else None
    assert gemm_add_layernorm_small_ok(a2d, w, 256, B, L), "gemm_add_layernorm_small: decoder-sized rows only (see gemm_add_layernorm_ok)"
    with _timed("gemm_add_layernorm_small[%dx256x%d]" % (M, K), 2.0 * M * 256 * K):
        check(lib().asr_gemm_add_layernorm_small(_stream(), _p(a2d), K, _p(w), _p(bias), _p(residual), _p(gamma), _p(beta), _p(row_len),
                                                 _p(s_sum), _p(y32), _p(y16), _p(mean), _p(rstd), B, L, K, float(eps), _d(drop_x)),
              "asr_gemm_add_layernorm_small")
    return s_sum, y32, y16, mean, rstd


gemm_add_layernorm = gemm_add_layernorm_small


PROJ_LN = True           # the attention output projection + residual + LayerNorm as one launch at encoder size (+8 us per layer as two)


def proj_ln_ok(a2d, w, D, B, L):
    """Shapes asr_proj_ln_fwd takes: encoder-sized rows (the decoder's go through gemm_add_layernorm_small), 256 -> 256."""
    return (PROJ_LN and B * L >= PROJ_LN_MIN_ROWS and D == 256 and tuple(w.shape) == (256, 256) and a2d.dtype == torch.bfloat16 and
            w.dtype == torch.bfloat16 and a2d.shape[1] == 256 and a2d.is_contiguous() and w.is_contiguous() and B * L * 1024 < 2 ** 31)


def proj_ln(a2d, w, bias, residual, gamma, beta, B, L, row_len=None, want_bf16=True, eps=1e-5, save_stats=False, drop_x=None, save_s=True):
    """LayerNorm(dropout_x(A . W^T + bias) + residual) for encoder-sized rows in one launch (asr_hip.h: asr_proj_ln_fwd).
    -> (s_sum [M,256] pre-norm sum or None, y32, y16 or None, mean, rstd): the tensors of the gemm_nt + add_layernorm pair."""
    _req_cuda(a2d, w, bias, residual, gamma, beta, row_len)
    M = a2d.shape[0]
    assert M == B * L and residual.is_contiguous()
    dev = a2d.device
    s_sum = torch.empty((M, 256), device=dev, dtype=torch.float32) if (save_stats and save_s) else None
    y32 = torch.empty((M, 256), device=dev, dtype=torch.float32)
    y16 = torch.empty((M, 256), device=dev, dtype=torch.bfloat16) if want_bf16 else None
    mean = torch.empty(M, device=dev, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(M, device=dev, dtype=torch.float32) if save_stats else None
    with _timed("proj_ln[%dx256x256]" % M, 2.0 * M * 256 * 256):
        check(lib().asr_proj_ln_fwd(_stream(), _p(a2d), _p(residual), _p(w), _p(bias), _p(gamma), _p(beta), _p(row_len), _p(s_sum), _p(y32),
                                    _p(y16), _p(mean), _p(rstd), B, L, 256, float(eps), _d(drop_x)), "asr_proj_ln_fwd")
    return s_sum, y32, y16, mean, rstd


FUSED_FFN = True         # the one-launch feed-forward sub-layer (False: two GEMMs + LayerNorm / two data-gradient GEMMs, step +1.9 ms)
# below: a 128-token block per CU leaves most of the chip idle and the launch takes its ~75 us whatever the row count, while the GEMM +
# GEMM + LayerNorm path scales with the rows (114 us at 32000): the crossover is near 20000 rows (S2 / CIF_Model, 8000 rows: 6.67 ->
# 6.40 ms and 7.18 -> 6.71 ms per step on the separate launches)
FUSED_FFN_MIN_ROWS = 20000
PROJ_LN_MIN_ROWS = 4096


def ffn_fused_ok(x16, x32, w1, w2, B, L):
    """Shapes asr_ffn_fwd / asr_ffn_bwd take: d_model = 256, d_ff a multiple of 64 up to 2048, bf16 operands, encoder-sized rows."""
    return (FUSED_FFN and x16 is not None and x16.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and
            x32.shape[1] == 256 and w1.shape[1] == 256 and w2.shape[0] == 256 and w1.shape[0] % 64 == 0 and 64 <= w1.shape[0] <= 2048 and
            B * L >= FUSED_FFN_MIN_ROWS and B * L * w1.shape[0] * 2 < 2 ** 31 and x16.is_contiguous() and x32.is_contiguous() and
            w1.is_contiguous() and w2.is_contiguous())


def ffn_fwd(x16, x32, w1, b1, w2, b2, gamma, beta, B, L, row_len=None, eps=1e-5, train=False, drop_x=None, want_bf16=True, save_s=True):
    """The whole position-wise feed-forward sub-layer in one launch (asr_hip.h: asr_ffn_fwd).
    -> (hid bf16 [M,d_ff] or None, bits or None, s_sum [M,256] or None, y32, y16, mean, rstd); hid / bits / s_sum / mean / rstd with train."""
    _req_cuda(x16, x32, w1, b1, w2, b2, gamma, beta, row_len)
    M, dff, dev = B * L, w1.shape[0], x32.device
    hid = torch.empty((M, dff), device=dev, dtype=torch.bfloat16) if train else None
    bits = torch.empty((int(lib().asr_ffn_bits_words(M, dff)),), device=dev, dtype=torch.int32) if train else None
    s_sum = torch.empty((M, 256), device=dev, dtype=torch.float32) if (train and save_s) else None      # (save_s=False: the backward takes x^ from y32)
    y32 = torch.empty((M, 256), device=dev, dtype=torch.float32)
    y16 = torch.empty((M, 256), device=dev, dtype=torch.bfloat16) if want_bf16 else None
    mean = torch.empty(M, device=dev, dtype=torch.float32) if train else None
    rstd = torch.empty(M, device=dev, dtype=torch.float32) if train else None
    with _timed("ffn_fwd[%dx256x%d]" % (M, dff), 4.0 * M * 256 * dff):
        check(lib().asr_ffn_fwd(_stream(), _p(x16), _p(x32), _p(w1), _p(b1), _p(w2), _p(b2), _p(gamma), _p(beta), _p(row_len), _p(hid),
                                _p(bits), _p(s_sum), _p(y32), _p(y16), _p(mean), _p(rstd), B, L, 256, dff, float(eps), _d(drop_x)),
              "asr_ffn_fwd")
    return hid, bits, s_sum, y32, y16, mean, rstd


ATTN_FFN_FUSED = os.environ.get("ASR_AMD_ATTN_FFN", "1") != "0"      # the attention sub-layer's tail and the feed-forward sub-layer of an encoder layer as ONE launch (asr_attn_ffn_fwd)


def attn_ffn_ok(ctx2d, wo, x32, w1, w2, B, L):
    """Shapes asr_attn_ffn_fwd takes: those of proj_ln AND of the fused feed-forward sub-layer."""
    return (ATTN_FFN_FUSED and proj_ln_ok(ctx2d, wo, x32.shape[1], B, L) and
            ffn_fused_ok(ctx2d, x32, w1, w2, B, L))


def attn_ffn_fwd(ctx2d, wo, bo, residual, gamma0, beta0, w1, b1, w2, b2, gamma, beta, B, L, row_len=None, eps0=1e-5, eps=1e-5, train=False,
                 drop0=None, drop_x=None, save_s=True):
    """proj_ln + ffn_fwd in one launch (asr_hip.h: asr_attn_ffn_fwd) -> ((s0 or None, x32, x16, mean0, rstd0), ffn_fwd's tuple): the tensors of
    the two calls it replaces, bit for bit."""
    _req_cuda(ctx2d, wo, bo, residual, gamma0, beta0, w1, b1, w2, b2, gamma, beta, row_len)
    M, dff, dev = B * L, w1.shape[0], residual.device
    assert ctx2d.shape[0] == M and residual.is_contiguous()
    f32, b16 = torch.float32, torch.bfloat16
    s0 = torch.empty((M, 256), device=dev, dtype=f32) if (train and save_s) else None
    x32 = torch.empty((M, 256), device=dev, dtype=f32)
    x16 = torch.empty((M, 256), device=dev, dtype=b16)
    mean0 = torch.empty(M, device=dev, dtype=f32) if train else None
    rstd0 = torch.empty(M, device=dev, dtype=f32) if train else None
    hid = torch.empty((M, dff), device=dev, dtype=b16) if train else None
    bits = torch.empty((int(lib().asr_ffn_bits_words(M, dff)),), device=dev, dtype=torch.int32) if train else None
    s_sum = torch.empty((M, 256), device=dev, dtype=f32) if (train and save_s) else None
    y32 = torch.empty((M, 256), device=dev, dtype=f32)
    y16 = torch.empty((M, 256), device=dev, dtype=b16)
    mean = torch.empty(M, device=dev, dtype=f32) if train else None
    rstd = torch.empty(M, device=dev, dtype=f32) if train else None
    with _timed("attn_ffn_fwd[%dx256x%d]" % (M, dff), 4.0 * M * 256 * dff + 2.0 * M * 256 * 256):
        check(lib().asr_attn_ffn_fwd(_stream(), _p(ctx2d), _p(residual), _p(wo), _p(bo), _p(gamma0), _p(beta0), float(eps0), _d(drop0), _p(s0),
                                     _p(x32), _p(x16), _p(mean0), _p(rstd0), _p(w1), _p(b1), _p(w2), _p(b2), _p(gamma), _p(beta), _p(row_len),
                                     _p(hid), _p(bits), _p(s_sum), _p(y32), _p(y16), _p(mean), _p(rstd), B, L, 256, dff, float(eps), _d(drop_x)),
              "asr_attn_ffn_fwd")
    return (s0, x32, x16, mean0, rstd0), (hid, bits, s_sum, y32, y16, mean, rstd)


def ffn_bwd(ds16, ds32, w1, w2, bits):
    """(d_hid bf16 [M,d_ff], dx f32 [M,256]) = the sub-layer's data gradient in one launch (asr_hip.h: asr_ffn_bwd)."""
    _req_cuda(ds16, ds32, w1, w2, bits)
    M, dff = ds32.shape[0], w1.shape[0]
    assert ds16.dtype == torch.bfloat16 and ds16.is_contiguous() and ds32.is_contiguous()
    d_hid = torch.empty((M, dff), device=ds32.device, dtype=torch.bfloat16)
    dx = torch.empty((M, 256), device=ds32.device, dtype=torch.float32)
    with _timed("ffn_bwd[%dx256x%d]" % (M, dff), 4.0 * M * 256 * dff):
        check(lib().asr_ffn_bwd(_stream(), _p(ds16), _p(ds32), _p(w1), _p(w2), _p(bits), _p(d_hid), _p(dx), M, 256, dff), "asr_ffn_bwd")
    return d_hid, dx


def ffn_bwd_ln(ds16, ds32, w1, w2, bits, B, L, ln_s, ln_mean, ln_rstd, ln_gamma, row_len, dgamma, dbeta, dbias=None, drop_x=None,
               ln_beta=None):
    """ffn_bwd whose dx goes straight through the backward of the LayerNorm that produced the sub-layer's input (asr_ffn_bwd_ln):
    -> (d_hid bf16 [M,d_ff], ds f32 [M,256], ds16 bf16 [M,256]) with ds / ds16 as add_layernorm_bwd(dy=dx, ...) returns them;
    dgamma / dbeta / dbias accumulated in place."""
    _req_cuda(ds16, ds32, w1, w2, bits, ln_s, ln_rstd, ln_gamma, dgamma, dbeta)
    M, dff = ds32.shape[0], w1.shape[0]
    assert M == B * L and ds16.dtype == torch.bfloat16 and ds16.is_contiguous() and ds32.is_contiguous() and ln_s.is_contiguous()
    d_hid = torch.empty((M, dff), device=ds32.device, dtype=torch.bfloat16)
    ds = torch.empty((M, 256), device=ds32.device, dtype=torch.float32)
    ds_b = torch.empty((M, 256), device=ds32.device, dtype=torch.bfloat16)
    with _timed("ffn_bwd[%dx256x%d]" % (M, dff), 4.0 * M * 256 * dff):
        check(lib().asr_ffn_bwd_ln(_stream(), _p(ds16), _p(ds32), _p(w1), _p(w2), _p(bits), _p(d_hid), B, L, 256, dff, _p(ln_s), _p(ln_mean),
                                   _p(ln_rstd), _p(ln_gamma), _p(ln_beta), _p(row_len), _p(ds), _p(ds_b), _p(dgamma), _p(dbeta), _p(dbias),
                                   _d(drop_x)),
              "asr_ffn_bwd_ln")
    return d_hid, ds, ds_b


def add_layernorm(x, residual, gamma, beta, B, L, pe=None, row_len=None, want_bf16=False, eps=1e-5, save_stats=False,
                  drop_x=None, drop_y=None):
    """y = LN(x [+ residual]) [+ pe[t]] [masked to t < row_len[b]] -> (y32 [B*L,D], y16 or None, mean, rstd).
    With save_stats the pre-norm sum x+residual overwrites x in place (it is what the backward needs)."""
    _req_cuda(x, residual, gamma, beta, pe, row_len)
    D = x.shape[-1]
    assert x.is_contiguous() and x.numel() == B * L * D
    y32 = torch.empty((B * L, D), device=x.device, dtype=torch.float32)
    y16 = torch.empty((B * L, D), device=x.device, dtype=torch.bfloat16) if want_bf16 else None
    mean = torch.empty(B * L, device=x.device, dtype=torch.float32) if save_stats else None
    rstd = torch.empty(B * L, device=x.device, dtype=torch.float32) if save_stats else None
    # algorithmic bytes: x and the residual read, y32 (+ the bf16 shadow) written, + the pre-norm sum written back over x when
    # the backward needs it (save_stats)
    nbytes = B * L * D * (4 + (4 if residual is not None else 0) + 4 + (2 if want_bf16 else 0) + (4 if save_stats else 0))
    with _timed("add_layernorm[%dx%d]" % (B * L, D), float(nbytes)):
        check(lib().asr_add_layernorm_fwd(_stream(), _p(x), _p(residual), _p(gamma), _p(beta), _p(pe), _p(row_len), _p(y32), _p(y16),
                                          _p(mean), _p(rstd), _p(x) if save_stats else None, B, L, D, float(eps), _d(drop_x),
                                          _d(drop_y)),
              "asr_add_layernorm_fwd")
    return y32, y16, mean, rstd


_ORDER_EVENTS = {"pool": [], "next": 0}
LIGHT_EVENTS = True      # cross-stream ordering on events without the system-scope fence (False: torch events, step +0.1 ms)


def order_after(later, earlier):
    """Everything queued on stream `earlier` so far happens before anything queued on stream `later` from now on (same device).
    An event without the system-scope fence of a default HIP event (asr_hip.h: asr_stream_order_after) from a ring of 256: a wait
    holds on to the record it was queued against, so an event can be recorded again while earlier waits on it are still queued."""
    if not LIGHT_EVENTS or torch.cuda.is_current_stream_capturing():
        ev = torch.cuda.Event()      # (under stream capture torch's own events: their lifetime is tied to the capture)
        ev.record(earlier)
        later.wait_event(ev)
        return
    pool = _ORDER_EVENTS["pool"]
    if len(pool) < 256:
        h = ctypes.c_void_p()
        check(lib().asr_event_create(ctypes.byref(h)), "asr_event_create")
        pool.append(h)
    i = _ORDER_EVENTS["next"] % len(pool)
    _ORDER_EVENTS["next"] += 1
    check(lib().asr_stream_order_after(ctypes.c_void_p(later.cuda_stream), ctypes.c_void_p(earlier.cuda_stream), pool[i]),
          "asr_stream_order_after")


def decoder_targets(targets, sos_id, eos_id, umax, overflow=None):
    """Decoder.preprocess (decoder.py:42-58) in one launch -> (ys_in [B, umax + 1], ys_out [B, umax + 1], in_len int32 [B]).
    `umax` = the longest target (non-pad entries) of the batch; `overflow` (int32 [1], optional) is set if a row held more."""
    _req_cuda(targets)
    assert targets.dtype == torch.int64 and targets.dim() == 2
    targets = targets.contiguous()
    B, U = targets.shape
    W = int(umax) + 1
    ys_in = torch.empty((B, W), device=targets.device, dtype=torch.int64)
    ys_out = torch.empty((B, W), device=targets.device, dtype=torch.int64)
    in_len = torch.empty((B,), device=targets.device, dtype=torch.int32)
    check(lib().asr_decoder_targets(_stream(), _p(targets), _p(ys_in), _p(ys_out), _p(in_len), None, _p(overflow), B, U, W,
                                    int(sos_id), int(eos_id)), "asr_decoder_targets")
    return ys_in, ys_out, in_len


def decoder_cif_targets(target, sos_id):
    """Decoder_CIF.preprocess (decoder.py:356-366) + lengths in one launch -> (ys_in [B, U] int64, in_len int32 [B])."""
    _req_cuda(target)
    assert target.dtype == torch.int64 and target.dim() == 2
    target = target.contiguous()
    B, U = target.shape
    ys_in = torch.empty_like(target)
    in_len = torch.empty((B,), device=target.device, dtype=torch.int32)
    check(lib().asr_decoder_cif_targets(_stream(), _p(target), _p(ys_in), _p(in_len), B, U, int(sos_id)), "asr_decoder_cif_targets")
    return ys_in, in_len


def embed_pe(ids, emb, pe, want_bf16=False, drop=None):
    _req_cuda(ids, emb, pe)
    B, U = ids.shape
    V, D = emb.shape
    ids = ids.contiguous()
    y32 = torch.empty((B * U, D), device=emb.device, dtype=torch.float32)
    y16 = torch.empty((B * U, D), device=emb.device, dtype=torch.bfloat16) if want_bf16 else None
    check(lib().asr_embed_pe_fwd(_stream(), _p(ids), _p(emb), _p(pe), _p(y32), _p(y16), B, U, D, V, _d(drop)), "asr_embed_pe_fwd")
    return y32, y16


def conv_sub0(feats, w0, b0, dtype, T1, F1):
    _req_cuda(feats, w0, b0)
    B, T, D = feats.shape
    y = torch.empty((B, T1, F1, 32), device=feats.device, dtype=dtype)
    check(lib().asr_conv_sub0_fwd(_stream(), _p(feats.contiguous()), _p(w0), _p(b0), _p(y), dtype_code(y), B, T, D, T1, F1),
          "asr_conv_sub0_fwd")
    return y


def conv_sub1(x, w, b, Tout, Fout, last):
    _req_cuda(x, w, b)
    B, Tin, Fin, C = x.shape
    assert C == 32 and x.is_contiguous()
    shape = (B, Tout, 32 * Fout) if last else (B, Tout, Fout, 32)
    y = torch.empty(shape, device=x.device, dtype=x.dtype)
    check(lib().asr_conv_sub1_fwd(_stream(), _p(x), _p(w), _p(b), _p(y), dtype_code(x), B, Tin, Fin, Tout, Fout, 1 if last else 0),
          "asr_conv_sub1_fwd")
    return y


def cast_bf16(x):
    _req_cuda(x)
    x = x.contiguous()
    y = torch.empty(x.shape, device=x.device, dtype=torch.bfloat16)
    check(lib().asr_cast_f32_bf16(_stream(), _p(x), _p(y), x.numel()), "asr_cast_f32_bf16")
    return y


def mask_rows_(x, lens):
    B, L, V = x.shape
    assert x.stride(2) == 1 and x.stride(0) == L * x.stride(1)
    check(lib().asr_mask_rows(_stream(), _p(x), _p(lens), B, L, V, x.stride(1)), "asr_mask_rows")
    return x


def assigner_tail(x, w, b, lens, B, L):
    Dh = x.shape[-1]
    alpha = torch.empty((B, L), device=x.device, dtype=torch.float32)
    check(lib().asr_assigner_tail_fwd(_stream(), _p(x), _p(w), _p(b), _p(lens), B, L, Dh, _p(alpha)), "asr_assigner_tail_fwd")
    return alpha


# ------------------------------------------------------------------------------------------------------------
class CtcState:
    """Workspaces saved between asr_ctc_loss_fwd and asr_ctc_loss_bwd."""
    __slots__ = ("logits", "ldl", "in_len", "targets", "B", "L", "V", "Umax", "blank", "lse", "lp_ext", "alpha", "nll",
                 "tgt_len")


_AUX = {}
# chunks of the fused CTC forward (asr_hip.h: asr_ctc_loss_fwd n_chunks): the hand-off granularity between the streaming pass and the
# recursion wavefronts of the same launch.  <= 1 = the two-launch form (pass, then recursion): 154-167 us at the north-star
# shape against 133-139 fused with 32 chunks
CTC_CHUNKS = 32
_CTC_COUNTERS = {}
_CTC_DBG = int(os.environ.get("ASR_AMD_CTC_DBG", "0") or 0) & 0xff


def ctc_reset_counters():
    """Forget the cached arrival-counter buffers of the one-launch CTC forward.  The launch hands its counters back zeroed when it ends
    cleanly; after anything else - a NaN loss from its bounded wait (a producer that never arrived), a device fault, an aborted
    stream - a caller that reads the loss back and finds it non-finite calls this, so that the next call starts from zeroed words instead of passing its
    gates early on stale arrivals."""
    _CTC_COUNTERS.clear()

CTC_LAZY_OCC = True     # asr_ctc_loss_bwd with the second workspace (see asr_hip.h)


QUEUE_PROBE = True      # side streams picked by hardware-queue probes (False: as torch hands them out - S1 +1 ms, S2 +3 ms in a quarter to three quarters of the runs)


def streams_share_queue(a, b):
    """Set-up time probe (asr_streams_share_queue): True when the two torch streams sit on one hardware queue (no overlap between them)."""
    out = ctypes.c_int(0)
    check(lib().asr_streams_share_queue(ctypes.c_void_p(a.cuda_stream), ctypes.c_void_p(b.cuda_stream), ctypes.byref(out)),
          "asr_streams_share_queue")
    return bool(out.value)


def aux_stream(device, priority=0, slot=0):
    """A side stream per (device, priority, slot): -1 for latency-bound work beside an HBM-bound pass (pipelined CTC forward), 0 for
    bulk work that should fill the CUs a run of small kernels leaves idle (the trainer's CTC branch beside the decoder; slot 1:
    the encoder's dropout-mask hashing beside its GEMMs).  The runtime multiplexes streams onto 4 hardware queues and two streams of one
    queue never overlap, so a new side stream is chosen among a few candidates: one that shares its queue neither with the current
    (launch) stream nor, if possible, with the side streams already handed out."""
    key = (torch.device(device).index, priority, slot)
    if key not in _AUX:
        cand = [torch.cuda.Stream(device=device, priority=priority)]
        if QUEUE_PROBE and torch.device(device).type == "cuda" and not torch.cuda.is_current_stream_capturing():
            main = torch.cuda.current_stream(device)
            taken = [s for k, s in _AUX.items() if k[0] == key[0]]
            cand += [torch.cuda.Stream(device=device, priority=priority) for _ in range(7)]
            free = [c for c in cand if not streams_share_queue(main, c)]
            alone = [c for c in free if not any(streams_share_queue(t, c) for t in taken)]
            cand = alone or free or cand
        _AUX[key] = cand[0]
    return _AUX[key]


FUSED_VOCAB_CTC = True      # the trainer's CTC branch: ctc_fc writing fp16 logits + lse + the CTC table rows in one launch (False: plain GEMM + the streaming CTC forward)


def vocab_proj_ctc_ok(x16, w16, B, L, Umax):
    """shapes asr_vocab_proj_ctc takes (the caller decides from which row count it is worth it)"""
    V = w16.shape[0]
    return (x16 is not None and x16.dtype == torch.bfloat16 and w16.dtype == torch.bfloat16 and x16.dim() == 2 and x16.shape[1] == 256 and
            w16.shape[1] == 256 and x16.is_contiguous() and w16.is_contiguous() and L >= 128 and 1 <= Umax and Umax + 1 <= 64 and
            x16.shape[0] == B * L and B * L < (1 << 24) and B * L * _pad8(V) * 2 < 2 ** 31 and V < (1 << 22))


def vocab_proj_ctc(x16, w16, targets, in_len, B, L, blank=None):
    """The training step's CTC branch forward in two launches (asr_hip.h: asr_vocab_proj_ctc + asr_ctc_loss_fwd_table): the projection
    writes fp16 logits, the rows' log-sum-exp and the CTC table rows; the alpha / beta recursion runs on the table.
    -> (logits fp16 [B*L, V] view of rows padded to 8, loss [1], nll [B], state for ctc_loss_bwd)"""
    _req_cuda(x16, w16, targets)
    M, V = x16.shape[0], w16.shape[0]
    Umax = targets.shape[1]
    Vp = _pad8(V)
    dev = x16.device
    st = CtcState()
    st.targets = targets.to(torch.int64).contiguous()
    st.in_len = as_i32(in_len)
    st.blank = V - 1 if blank is None else blank
    buf = torch.empty((M, Vp), device=dev, dtype=torch.float16)
    st.logits, st.ldl, st.B, st.L, st.V, st.Umax = buf.view(B, L, Vp)[:, :, :V], Vp, B, L, V, Umax
    S = lib().asr_ctc_workspace_stride(Umax)
    st.lse = torch.empty((B, L), device=dev, dtype=torch.float32)
    st.lp_ext = torch.empty((B, L, S), device=dev, dtype=torch.float32)
    st.alpha = torch.empty((B, L + 2, S), device=dev, dtype=torch.float32)
    st.nll = torch.empty(B, device=dev, dtype=torch.float32)
    st.tgt_len = torch.empty(B, device=dev, dtype=torch.int32)
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    with _timed("vocab_proj_ctc[%dx%dx256]" % (M, V), 2.0 * M * V * 256):
        check(lib().asr_vocab_proj_ctc(_stream(), _p(x16), _p(w16), _p(buf), Vp, _p(st.lse), _p(st.lp_ext), _p(st.targets), B, L, V, Umax,
                                       st.blank, 256), "asr_vocab_proj_ctc")
    # (the table form touches the table only: lp_ext read, alpha written - NOT the logits; its own name so no table prices it at 4 B L V)
    with _timed("ctc_loss_fwd_table[B%d L%d S%d U%d]" % (B, L, S, Umax), 2.0 * 4.0 * B * L * S):
        check(lib().asr_ctc_loss_fwd_table(_stream(), _p(st.lp_ext), _p(st.in_len), _p(st.targets), B, L, Umax, _p(st.alpha), _p(st.nll),
                                           _p(st.tgt_len), _p(loss)), "asr_ctc_loss_fwd_table")
    return buf[:, :V], loss, st.nll, st


def ctc_loss_fwd(logits, in_len, targets, blank=None, n_chunks=None):
    """logits f32 [B,L,V] (last dim contiguous, rows may be strided), in_len int32 [B], targets int64 [B,Umax].
    -> (loss scalar tensor [1], nll [B], state)"""
    _req_cuda(logits, in_len, targets)
    B, L, V = logits.shape
    assert logits.dtype == torch.float32 and logits.stride(2) == 1 and logits.stride(0) == L * logits.stride(1)
    Umax = targets.shape[1]
    st = CtcState()
    st.logits, st.ldl, st.B, st.L, st.V, st.Umax = logits, logits.stride(1), B, L, V, Umax
    st.blank = V - 1 if blank is None else blank
    st.in_len = as_i32(in_len)
    st.targets = targets.to(torch.int64).contiguous()
    S = lib().asr_ctc_workspace_stride(Umax)   # opaque workspace row stride (asr_hip.h)
    dev = logits.device
    st.lse = torch.empty((B, L), device=dev, dtype=torch.float32)
    st.lp_ext = torch.empty((B, L, S), device=dev, dtype=torch.float32)
    st.alpha = torch.empty((B, L + 2, S), device=dev, dtype=torch.float32)   # +2 rows: beta at the meeting point, arrival counters
    st.nll = torch.empty(B, device=dev, dtype=torch.float32)
    st.tgt_len = torch.empty(B, device=dev, dtype=torch.int32)
    nck = CTC_CHUNKS if n_chunks is None else n_chunks
    # arrival counters of the fused form from the step's zero arena when one is live (no memset node in front of the launch)
    counters = None
    nwords = int(lib().asr_ctc_counter_words(B, L, nck)) if (_ARENA["live"] and _ARENA["buf"].device == dev and Umax + 1 <= 64) else 0
    if nwords:
        off = (_ARENA["off"] + 63) // 64 * 64
        if off + nwords <= _ARENA["buf"].numel():
            _ARENA["off"] = _ARENA["dirty"] = off + nwords
            counters = _ARENA["buf"][off:off + nwords]
    if counters is None and Umax + 1 <= 64 and not torch.cuda.is_current_stream_capturing():
        # outside a step's zero arena: one counter buffer per (device, stream), zeroed once - the kernel hands it back zeroed
        nw = int(lib().asr_ctc_counter_words(B, L, nck))
        if nw:
            key = (dev.index, _stream().value, nw)
            if _CTC_DBG:
                # a diagnostic build of the launch (ASR_AMD_CTC_DBG: pass only, chains only, ...) does not end by handing the counters
                # back zeroed: never let such a call share the cached buffer with a later one
                counters = torch.zeros(nw, device=dev, dtype=torch.int32)
            else:
                if key not in _CTC_COUNTERS:
                    _CTC_COUNTERS[key] = torch.zeros(nw, device=dev, dtype=torch.int32)
                counters = _CTC_COUNTERS[key]
    loss = torch.empty(1, device=dev, dtype=torch.float32)
    with _timed("ctc_loss_fwd[B%d L%d V%d U%d]" % (B, L, V, Umax), 4.0 * B * L * V):
        check(lib().asr_ctc_loss_mean_fwd(_stream(), _p(logits), st.ldl, _p(st.in_len), _p(st.targets), B, L, V, Umax, st.blank, _p(st.lse),
                                          _p(st.lp_ext), _p(st.alpha), _p(st.nll), _p(st.tgt_len),
                                          _p(counters), nck, _p(loss)), "asr_ctc_loss_mean_fwd")
    return loss, st.nll, st


def _pad8(v):
    return (v + 7) // 8 * 8


def ctc_loss_bwd(st, gout, bf16=False):
    """-> grad wrt logits as a [B,L,V] view of a zero-padded [B,L,Vp] buffer (rows 16-byte aligned so the gradient is directly a
    GEMM operand).  Consumes st.alpha.  bf16=True (a gradient that only feeds the projection's backward GEMMs, which run on bf16
    MFMA anyway): half the bytes, Vp = roundup(V, 128) so that both GEMMs take their LDS-DMA kernels, pad written by the kernel."""
    fp16_in = st.logits.dtype == torch.float16
    if fp16_in and not bf16:
        raise ValueError("ctc_loss_bwd: an fp16 logits image (asr_vocab_proj_ctc) only has the bf16 gradient form - pass bf16=True")
    if fp16_in or (bf16 and st.ldl % 4 == 0 and st.logits.data_ptr() % 16 == 0):
        Vp = (st.V + 127) // 128 * 128
        gbuf = torch.empty((st.B, st.L, Vp), device=st.logits.device, dtype=torch.bfloat16)
    else:
        Vp = _pad8(st.V)
        gbuf = torch.empty((st.B, st.L, Vp), device=st.logits.device, dtype=torch.float32)
        if Vp != st.V:
            gbuf[:, :, st.V:].zero_()      # only the pad columns (the kernel writes every real one): not a 542 MB fill at S1
    grad = gbuf[:, :, :st.V]
    gout = gout.reshape(1).to(torch.float32).contiguous()
    # second workspace: the recursion's second half stores raw rows, the gradient pass forms the occupancies (CTC_LAZY_OCC)
    alpha2 = torch.empty_like(st.alpha) if CTC_LAZY_OCC else None
    with _timed("ctc_loss_bwd[B%d L%d V%d U%d]" % (st.B, st.L, st.V, st.Umax), (4.0 + gbuf.element_size()) * st.B * st.L * st.V):
        check(lib().asr_ctc_loss_bwd_ex(_stream(), _p(st.logits), dtype_code(st.logits), st.ldl, _p(st.in_len), _p(st.targets), st.B, st.L, st.V,
                                        st.Umax, st.blank, _p(st.lse), _p(st.lp_ext), _p(st.alpha), _p(st.nll), _p(st.tgt_len), _p(gout),
                                        _p(grad), dtype_code(gbuf), Vp, _p(alpha2)), "asr_ctc_loss_bwd")
    return grad


def ce_loss_fwd(logits2d, targets1d, smoothing):
    """-> (loss[2] = (mean, n_word), row_loss [N], lse [N])"""
    _req_cuda(logits2d, targets1d)
    N, V = logits2d.shape
    assert logits2d.stride(1) == 1 and logits2d.dtype == torch.float32
    targets1d = targets1d.to(torch.int64).contiguous()
    row_loss = torch.empty(N, device=logits2d.device, dtype=torch.float32)
    lse = torch.empty(N, device=logits2d.device, dtype=torch.float32)
    check(lib().asr_ce_loss_fwd(_stream(), _p(logits2d), logits2d.stride(0), _p(targets1d), N, V, float(smoothing), _p(row_loss),
                                _p(lse)), "asr_ce_loss_fwd")
    loss = torch.empty(2, device=logits2d.device, dtype=torch.float32)
    check(lib().asr_ce_mean(_stream(), _p(row_loss), _p(targets1d), N, _p(loss)), "asr_ce_mean")
    return loss, row_loss, lse, targets1d


def ce_mask_loss_fwd(logits2d, targets1d, counted1d, smoothing):
    """mask_lm's cal_ce_mask_loss: like ce_loss_fwd, the denominator counting only rows with counted != 0 (bool / uint8 [N])."""
    _req_cuda(logits2d, targets1d, counted1d)
    N, V = logits2d.shape
    assert logits2d.stride(1) == 1 and logits2d.dtype == torch.float32
    targets1d = targets1d.to(torch.int64).contiguous()
    counted = counted1d.to(torch.uint8).contiguous()
    row_loss = torch.empty(N, device=logits2d.device, dtype=torch.float32)
    lse = torch.empty(N, device=logits2d.device, dtype=torch.float32)
    check(lib().asr_ce_loss_fwd(_stream(), _p(logits2d), logits2d.stride(0), _p(targets1d), N, V, float(smoothing), _p(row_loss),
                                _p(lse)), "asr_ce_loss_fwd")
    loss = torch.empty(2, device=logits2d.device, dtype=torch.float32)
    check(lib().asr_ce_mean_masked(_stream(), _p(row_loss), _p(targets1d), _p(counted), N, _p(loss)), "asr_ce_mean_masked")
    return loss, row_loss, lse, targets1d


def token_mask(ids, rand01, p=0.05, M=10):
    """Mask_LM.token_mask: -> (masked ids int64 [B,T], masked positions bool [B,T])"""
    _req_cuda(ids, rand01)
    ids = ids.to(torch.int64).contiguous()
    B, T = ids.shape
    rand01 = rand01.to(device=ids.device, dtype=torch.float32).contiguous()
    assert rand01.shape == (B, T)
    out = torch.empty_like(ids)
    masked = torch.empty((B, T), device=ids.device, dtype=torch.uint8)
    check(lib().asr_token_mask(_stream(), _p(ids), _p(rand01), B, T, float(p), int(M), _p(out), _p(masked)), "asr_token_mask")
    return out, masked.bool()


def ce_loss_bwd(logits2d, targets1d, smoothing, lse, loss2, gout, bf16=False):
    """bf16=True: the trainer's gradient image (see ctc_loss_bwd)."""
    N, V = logits2d.shape
    if bf16:
        Vp = (V + 127) // 128 * 128
        gbuf = torch.empty((N, Vp), device=logits2d.device, dtype=torch.bfloat16)
    else:
        Vp = _pad8(V)
        gbuf = torch.empty((N, Vp), device=logits2d.device, dtype=torch.float32)
        if Vp != V:
            gbuf[:, V:].zero_()
    grad = gbuf[:, :V]
    gout = gout.reshape(1).to(torch.float32).contiguous()
    n_word = loss2[1:2]
    check(lib().asr_ce_loss_bwd(_stream(), _p(logits2d), logits2d.stride(0), _p(targets1d), N, V, float(smoothing), _p(lse),
                                _p(n_word), _p(gout), _p(grad), dtype_code(gbuf), Vp), "asr_ce_loss_bwd")
    return grad


def cif_scan(alpha, threshold, want_tok=False):
    """alpha f32 [B,L] -> cur, rem [B,L], fire_idx int32 [B,L], n_fire [B], n_label [B] (+ tok int32 [B,L] when want_tok)"""
    _req_cuda(alpha)
    alpha = alpha.contiguous()
    B, L = alpha.shape
    dev = alpha.device
    cur = torch.empty((B, L), device=dev, dtype=torch.float32)
    rem = torch.empty((B, L), device=dev, dtype=torch.float32)
    fire_idx = torch.zeros((B, L), device=dev, dtype=torch.int32)
    n_fire = torch.empty(B, device=dev, dtype=torch.int32)
    n_label = torch.empty(B, device=dev, dtype=torch.int32)
    tok = torch.empty((B, L), device=dev, dtype=torch.int32) if want_tok else None
    check(lib().asr_cif_scan_fwd(_stream(), _p(alpha), B, L, float(threshold), _p(cur), _p(rem), _p(fire_idx), _p(n_fire), _p(n_label),
                                 _p(tok)), "asr_cif_scan_fwd")
    if want_tok:
        return cur, rem, fire_idx, n_fire, n_label, tok
    return cur, rem, fire_idx, n_fire, n_label


def cif_gather(hidden, cur, rem, fire_idx, n_fire, Umax):
    _req_cuda(hidden)
    hidden = hidden.contiguous()
    B, L, H = hidden.shape
    out = torch.empty((B, Umax, H), device=hidden.device, dtype=torch.float32)
    check(lib().asr_cif_gather_fwd(_stream(), _p(hidden), _p(cur), _p(rem), _p(fire_idx), _p(n_fire), B, L, H, Umax, _p(out)),
          "asr_cif_gather_fwd")
    return out


# ------------------------------------------------------------------------------------------------------------
# backward-pass ops
def relu_bits_buffer(M, N, device):
    """Workspace for gemm_nt_ex(relu_bits_out=) / gemm_nn(relu_bits=): roundup(M, 128) * N/8 bytes in the epilogues' own order
    (asr_hip.h), shaped [roundup(M, 128), N/8]."""
    return torch.empty(((M + 127) // 128 * 128, N // 8), device=device, dtype=torch.uint8)


def gemm_nt_ex(a2d, w, bias=None, out_dtype=torch.float32, relu=False, addend=None, relu_mask=None, out=None, relu_bits_out=None):
    """C = A . W^T (+bias) (+addend) (masked by relu_mask > 0).  A [M,K] contiguous; W [N,K].
    relu_bits_out: relu_bits_buffer(M, N) receiving the sign bits of the (ReLU'd, bf16) output - see asr_hip.h."""
    _req_cuda(a2d, w, bias, addend, relu_mask, relu_bits_out)
    M, K = a2d.shape
    N = w.shape[0]
    assert relu_bits_out is None or (relu_bits_out.is_contiguous() and relu_bits_out.numel() >= (M + 127) // 128 * 128 * (N // 8))
    assert a2d.is_contiguous() and w.is_contiguous() and w.shape[1] == K
    if out is None:
        out = torch.empty((M, N), device=a2d.device, dtype=out_dtype)
    with _timed("gemm_nt[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
        check(lib().asr_gemm_nt_ex(_stream(), _p(a2d), dtype_code(a2d), K, _p(w), dtype_code(w), K, _p(bias), _p(out), dtype_code(out),
                                   N, M, N, K, GEMM_RELU if relu else 0, _p(addend), N, _p(relu_mask), N, _p(relu_bits_out),
                                   relu_bits_out.stride(0) if relu_bits_out is not None else 0), "asr_gemm_nt_ex")
    return out


DGRAD_ROWS = True      # the row-block data gradient (False: the tiled GEMM for every data gradient)
DGRAD_ROWS_MIN = 8192
DGRAD_ROWS_MIN_K = 512      # (K = 256, the attention output projection: 20.4 against the tiled GEMM's 18.6 us)


def dgrad_rows_ok(a2d, w, K=None, lda=None):
    """Shapes asr_dgrad_rows is used for: bf16 dY [M >= DGRAD_ROWS_MIN, K % 64 == 0, K >= DGRAD_ROWS_MIN_K] (row stride % 8 == 0) against
    W bf16 [K, 256]."""
    K = a2d.shape[1] if K is None else K
    lda = a2d.stride(0) if lda is None else lda
    return (DGRAD_ROWS and a2d.is_cuda and a2d.dtype == torch.bfloat16 and w.dtype == torch.bfloat16 and w.dim() == 2 and w.shape[1] == 256 and
            w.is_contiguous() and w.shape[0] == K and K % 64 == 0 and K >= DGRAD_ROWS_MIN_K and a2d.shape[0] >= DGRAD_ROWS_MIN and lda % 8 == 0 and a2d.stride(1) == 1 and
            a2d.data_ptr() % 16 == 0)


def gemm_nn_ln(a2d, w, addend, B, L, ln_s, ln_mean, ln_rstd, ln_gamma, row_len, dgamma, dbeta, dbias=None, drop_x=None, ln_beta=None):
    """gemm_nn(a2d, w, addend=addend) followed by add_layernorm_bwd on its result, in one launch (asr_dgrad_rows_ln; dgrad_rows_ok
    shapes): -> (ds f32 [M,256], ds16 bf16 [M,256]); dgamma / dbeta / dbias accumulated in place."""
    _req_cuda(a2d, w, addend, ln_s, ln_rstd, ln_gamma, dgamma, dbeta)
    M, K = a2d.shape
    assert M == B * L and dgrad_rows_ok(a2d, w) and ln_s.is_contiguous() and (addend is None or addend.is_contiguous())
    ds = torch.empty((M, 256), device=a2d.device, dtype=torch.float32)
    ds_b = torch.empty((M, 256), device=a2d.device, dtype=torch.bfloat16)
    with _timed("gemm_nn[%dx256x%d]" % (M, K), 2.0 * M * 256 * K):
        check(lib().asr_dgrad_rows_ln(_stream(), _p(a2d), a2d.stride(0), _p(w), _p(addend), B, L, K, 256, _p(ln_s), _p(ln_mean), _p(ln_rstd),
                                      _p(ln_gamma), _p(ln_beta), _p(row_len), _p(ds), _p(ds_b), _p(dgamma), _p(dbeta), _p(dbias), _d(drop_x)),
              "asr_dgrad_rows_ln")
    return ds, ds_b


def gemm_nn(a2d, w, out_dtype=torch.float32, addend=None, relu_mask=None, lda=None, K=None, relu_bits=None):
    """dX[M,in] = A[M,out] . W[out,in]  (W bf16 as stored).  Optional f32 addend [M,in] and bf16 relu_mask [M,in] - or, instead
    of the latter, relu_bits uint8 [M, in/8] (the sign-bit image gemm_nt_ex wrote).
    `K` (<= a2d.shape[1], rows of W used) and `lda` let A live in a wider / padded buffer."""
    _req_cuda(a2d, w, addend, relu_mask, relu_bits)
    assert relu_mask is None or relu_bits is None
    M = a2d.shape[0]
    assert relu_bits is None or (relu_bits.is_contiguous() and relu_bits.numel() >= (M + 127) // 128 * 128 * (w.shape[1] // 8))
    K = a2d.shape[1] if K is None else K
    lda = a2d.stride(0) if lda is None else lda
    N = w.shape[1]
    if w.dtype == torch.float32:        # fp32 parity mode: dY . W as the exact-f32 NT GEMM on a transposed copy of W (test plumbing)
        assert relu_bits is None and lda == a2d.stride(0)
        Kp = (K + 3) // 4 * 4                                   # the GEMM wants K % 4 == 0: zero columns add nothing
        a = torch.zeros((M, Kp), device=a2d.device, dtype=torch.float32)
        a[:, :K].copy_(a2d[:, :K])
        wt = torch.zeros((N, Kp), device=a2d.device, dtype=torch.float32)
        wt[:, :K].copy_(w[:K].t())
        out = gemm_nt_ex(a, wt, None, torch.float32, addend=addend)
        if relu_mask is not None:
            out = relu_mask_mul(out, relu_mask.contiguous(), out=out)
        return out if out_dtype == torch.float32 else out.to(out_dtype)
    assert w.is_contiguous() and w.dtype == torch.bfloat16 and w.shape[0] >= K and a2d.stride(1) == 1
    if dgrad_rows_ok(a2d, w, K, lda) and relu_mask is None and relu_bits is None and out_dtype in (torch.float32, torch.bfloat16):
        # encoder-sized rows into d_model = 256: the row-block kernel (asr_dgrad_rows) - whole rows per workgroup, W streamed through LDS
        out = torch.empty((M, N), device=a2d.device, dtype=out_dtype)
        with _timed("gemm_nn[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
            check(lib().asr_dgrad_rows(_stream(), _p(a2d), lda, _p(w), _p(addend), _p(out), dtype_code(out), M, K, N), "asr_dgrad_rows")
        return out
    out = _arena_take(M, N, K, a2d.device) if (out_dtype == torch.float32 and relu_mask is None and relu_bits is None) else None
    zero_flag = 2 if out is not None else 0
    if out is None:
        out = torch.empty((M, N), device=a2d.device, dtype=out_dtype)
    with _timed("gemm_nn[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
        check(lib().asr_gemm_nn(_stream(), _p(a2d), dtype_code(a2d), lda, _p(w), N, None, _p(out), dtype_code(out), N, M, N, K,
                                _p(addend), N, _p(relu_bits if relu_bits is not None else relu_mask),
                                relu_bits.stride(0) if relu_bits is not None else N, (1 if relu_bits is not None else 0) | zero_flag), "asr_gemm_nn")
    return out


EXACT_F32 = False      # set by modules.set_precision("f32"): weight gradients on the exact-f32 MFMA GEMM instead of bf16 operands


def _gemm_tn_f32(a2d, b2d, out, accumulate, colsum_out):
    """fp32 parity mode: dW = A^T . B as the exact-f32 NT GEMM over transposed, zero-padded copies (test plumbing, not a fast path)"""
    M, N = a2d.shape
    K = b2d.shape[1]
    Mp = (M + 3) // 4 * 4
    at = torch.zeros((N, Mp), device=a2d.device, dtype=torch.float32)
    at[:, :M].copy_(a2d.t())
    bt = torch.zeros((K, Mp), device=a2d.device, dtype=torch.float32)
    bt[:, :M].copy_(b2d.t())
    res = gemm_nt_ex(at, bt, None, torch.float32)
    if out is None:
        out = res
    elif accumulate:
        add_(out, res)
    else:
        out.copy_(res)
    if colsum_out is not None:
        colsum(a2d, out=colsum_out, accumulate=True)
    return out


class _BudgetTLS(threading.local):      # per host thread, like the C side's thread_local (asr_launch_budget): one value drives both halves
    cus = 0


_BUDGET = _BudgetTLS()


class launch_budget:
    """`with ops.launch_budget(cus):` - launches queued inside keep to `cus` CUs' worth of the chip (asr_hip.h: asr_launch_budget; the
    weight-gradient GEMM takes it as its max_wgs).  For side-stream work beside a chain of small kernels; 0 = no budget."""

    def __init__(self, cus):
        self.cus = int(cus or 0)

    def __enter__(self):
        self.old_c, self.old_p = int(lib().asr_launch_budget(self.cus)), _BUDGET.cus
        _BUDGET.cus = self.cus
        return self

    def __exit__(self, *exc):
        lib().asr_launch_budget(self.old_c)
        _BUDGET.cus = self.old_p
        return False


def set_deterministic(on):
    """asr_set_deterministic: single-writer forms instead of float atomics in arrival order (forward split-K GEMMs, bias gradients)."""
    global DETERMINISTIC
    DETERMINISTIC = bool(on)
    return bool(lib().asr_set_deterministic(1 if on else 0))


def poison_lds(device="cuda"):
    """Test support (asr_debug_poison_lds): fill every CU's LDS with NaN patterns on the current stream."""
    scratch = torch.zeros(1, device=device, dtype=torch.int32)
    check(lib().asr_debug_poison_lds(_stream(), _p(scratch)), "asr_debug_poison_lds")
    return scratch


# Workspaces of the slab-reduced weight gradient (csrc/wgrad.hip): one per destination buffer, so that launches queued on different
# streams never share one, address-stable for graph capture.
TN_SLAB = True      # the slab-reduced weight gradient (False: the float-atomics kernel it replaced; tests/test_gpu_wgrad.py compares the two)
DETERMINISTIC = os.environ.get("ASR_AMD_DETERMINISTIC", "0") not in ("", "0")
_tn_ws = {}


def _tn_workspace(out, M, N, K, max_wgs, persistent=True):
    """persistent: `out` is a destination the caller keeps (a weight.grad view): its workspace is cached under its address.  A result
    gemm_tn allocated itself gets a workspace of its own that lives as long as the launch's stream order needs it (the caching
    allocator's) - cached under a fresh address every call it would never be evicted."""
    need = int(lib().asr_gemm_tn_ws_bytes(M, N, K, int(max_wgs)))
    if need <= 0:
        return None
    if not persistent:
        return torch.empty(need, device=out.device, dtype=torch.uint8)
    key = (out.data_ptr(), out.device.index)
    ws = _tn_ws.get(key)
    if ws is None or ws.numel() < need:
        ws = torch.empty(need, device=out.device, dtype=torch.uint8)
        _tn_ws[key] = ws
    return ws


def gemm_tn(a2d, b2d, out=None, accumulate=False, colsum=None, max_wgs=0):
    """dW[N,K] = A[M,N]^T . B[M,K]  (f32 result).  A/B f32 or bf16; rows may be strided (padded buffers).
    colsum (f32 [N]) += column sums of A (the bias gradient) in the same pass."""
    max_wgs = max_wgs or _BUDGET.cus
    _req_cuda(a2d, b2d)
    M, N = a2d.shape
    K = b2d.shape[1]
    assert a2d.stride(1) == 1 and b2d.stride(1) == 1 and b2d.shape[0] == M
    if EXACT_F32:
        return _gemm_tn_f32(a2d, b2d, out, accumulate, colsum)
    own_out = out is None
    if own_out:
        out = torch.empty((N, K), device=a2d.device, dtype=torch.float32)
        accumulate = False
    if TN_SLAB and a2d.dtype == torch.bfloat16 and b2d.dtype == torch.bfloat16 and K % 128 == 0 and M >= 64:
        ws = _tn_workspace(out, M, N, K, max_wgs, persistent=not own_out)
        if ws is not None:
            with _timed("gemm_tn[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
                check(lib().asr_gemm_tn_ws(_stream(), _p(a2d), a2d.stride(0), _p(b2d), b2d.stride(0), _p(out), out.stride(0), M, N, K,
                                           1 if accumulate else 0, _p(colsum), int(max_wgs), _p(ws), ws.numel(),
                                           1 if DETERMINISTIC else 0), "asr_gemm_tn_ws")
            return out
    with _timed("gemm_tn[%dx%dx%d]" % (M, N, K), 2.0 * M * N * K):
        check(lib().asr_gemm_tn(_stream(), _p(a2d), dtype_code(a2d), a2d.stride(0), _p(b2d), dtype_code(b2d), b2d.stride(0), _p(out),
                                out.stride(0), M, N, K,
                                0 if accumulate else 1, _p(colsum), int(max_wgs)), "asr_gemm_tn")
    return out


class _TnProblem(ctypes.Structure):          # asr_hip.h: asr_tn_problem_t
    _fields_ = [("A", ctypes.c_void_p), ("lda", ctypes.c_int64), ("B", ctypes.c_void_p), ("ldb", ctypes.c_int64), ("C", ctypes.c_void_p),
                ("ldc", ctypes.c_int64), ("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("accumulate", ctypes.c_int),
                ("colsum", ctypes.c_void_p), ("workspace", ctypes.c_void_p), ("workspace_bytes", ctypes.c_int64)]


TN_GROUP = True      # the decoder's weight gradients eight per launch pair (False: one pair each, +0.1 ms)
TN_GROUP_MAX_ROWS = 2048
TN_GROUP_SMALL_TILES = 16     # > 0: weight gradients of at most this many output tiles join the groups at any row count


def gemm_tn_group_ok(a2d, b2d, out):
    """A weight gradient the grouped launch takes: bf16 operands the slab kernel takes, decoder-sized row counts."""
    return (TN_GROUP and TN_SLAB and not EXACT_F32 and out is not None and a2d.dtype == torch.bfloat16 and b2d.dtype == torch.bfloat16 and
            (64 <= a2d.shape[0] <= TN_GROUP_MAX_ROWS or
             (TN_GROUP_SMALL_TILES and ((a2d.shape[1] + 127) // 128) * (b2d.shape[1] // 128) <= TN_GROUP_SMALL_TILES and a2d.shape[0] >= 64)) and
            b2d.shape[1] % 128 == 0 and a2d.stride(1) == 1 and b2d.stride(1) == 1 and
            a2d.stride(0) % 8 == 0 and b2d.stride(0) % 8 == 0 and a2d.data_ptr() % 16 == 0 and b2d.data_ptr() % 16 == 0 and
            (a2d.shape[1] % 128 == 0 or a2d.stride(0) >= (a2d.shape[1] + 127) // 128 * 128))


def tn_tiles(a2d, b2d):
    """128 x 128 output tiles of dW = a2d^T . b2d"""
    return ((a2d.shape[1] + 127) // 128) * (b2d.shape[1] // 128)


def gemm_tn_group(problems, group_wgs=0):
    """problems: up to 16 tuples (a2d, b2d, out, accumulate, colsum), each what asr_gemm_tn_ws takes - dW_i (+)= A_i^T . B_i in
    ONE pair of launches (asr_hip.h: asr_gemm_tn_ws_group_wgs).  group_wgs: the launch's workgroup budget (0: the library's default);
    <= the problems' total output tiles means every problem runs unsplit over M and there is no reduce launch."""
    arr = (_TnProblem * len(problems))()
    keep = []
    for i, (a2d, b2d, out, accumulate, colsum) in enumerate(problems):
        M, N = a2d.shape
        K = b2d.shape[1]
        ws = _tn_workspace(out, M, N, K, 0)
        keep.append(ws)
        arr[i] = _TnProblem(_p(a2d), a2d.stride(0), _p(b2d), b2d.stride(0), _p(out), out.stride(0), M, N, K, 1 if accumulate else 0,
                            _p(colsum), _p(ws), ws.numel())
    with _timed("gemm_tn_group[%d]" % len(problems), sum(2.0 * a.shape[0] * a.shape[1] * b.shape[1] for a, b, _, _, _ in problems)):
        check(lib().asr_gemm_tn_ws_group_wgs(_stream(), len(problems), ctypes.cast(arr, ctypes.c_void_p), 1 if DETERMINISTIC else 0,
                                             int(group_wgs)), "asr_gemm_tn_ws_group_wgs")


def colsum(a2d, out=None, accumulate=False):
    _req_cuda(a2d)
    M, N = a2d.shape
    assert a2d.stride(1) == 1
    if out is None:
        out = torch.empty(N, device=a2d.device, dtype=torch.float32)
        accumulate = False
    check(lib().asr_colsum(_stream(), _p(a2d), dtype_code(a2d), a2d.stride(0), M, N, _p(out), 0 if accumulate else 1), "asr_colsum")
    return out


def add_layernorm_bwd(dy, s, mean, rstd, gamma, row_len, B, L, dgamma, dbeta, want_bf16=False, dbias=None, drop_x=None,
                      drop_y=None, beta=None):
    """-> (ds f32 [M,D], ds16 or None); dgamma/dbeta (and dbias += colsum(ds) when given) accumulated in place.
    beta given: `s` is the LayerNorm's OUTPUT y (the forward kept no pre-norm sum; asr_add_layernorm_bwd_y), mean is not read."""
    _req_cuda(dy, s, rstd, gamma, row_len, dgamma, dbeta)
    D = s.shape[-1]
    assert dy.is_contiguous() and s.is_contiguous()
    ds = torch.empty((B * L, D), device=s.device, dtype=torch.float32)
    ds16 = torch.empty((B * L, D), device=s.device, dtype=torch.bfloat16) if want_bf16 else None
    nbytes = B * L * D * (4 + 4 + 4 + (2 if want_bf16 else 0))
    if beta is not None:
        assert drop_y is None
        with _timed("add_layernorm_bwd[%dx%d]" % (B * L, D), float(nbytes)):
            check(lib().asr_add_layernorm_bwd_y(_stream(), _p(dy), _p(s), _p(rstd), _p(gamma), _p(beta), _p(row_len), _p(ds), _p(ds16),
                                                _p(dgamma), _p(dbeta), _p(dbias), B, L, D, _d(drop_x)), "asr_add_layernorm_bwd_y")
        return ds, ds16
    with _timed("add_layernorm_bwd[%dx%d]" % (B * L, D), float(nbytes)):
        check(lib().asr_add_layernorm_bwd(_stream(), _p(dy), _p(s), _p(mean), _p(rstd), _p(gamma), _p(row_len), _p(ds), _p(ds16),
                                          _p(dgamma), _p(dbeta), _p(dbias), B, L, D, _d(drop_x), _d(drop_y)), "asr_add_layernorm_bwd")
    return ds, ds16


def attention_bwd(q, k, v, ctx, d_ctx, lse, k_len, causal, scale, dq_out, dk_out, dv_out, drop=None, drop_bits=None):
    """q [B,h,Lq,64], k/v [B,h,Lk,64] bf16; ctx, d_ctx token-major bf16 [B,Lq,h*64]; dq_out / dk_out / dv_out are bf16 views with
    row stride (elements) dq_out.stride(0) / dk_out.stride(0) into token-major gradient buffers (last dim = h*64)."""
    _req_cuda(q, k, v, ctx, d_ctx, lse)
    B, h, Lq, _ = q.shape
    Lk = k.shape[2]
    if q.dtype == torch.float32:        # fp32 parity mode: one slow VALU kernel, f32 gradients (no dropout in this mode)
        assert drop is None and ctx.is_contiguous() and d_ctx.is_contiguous() and d_ctx.dtype == torch.float32
        assert dq_out.dtype == dk_out.dtype == dv_out.dtype == torch.float32 and dk_out.stride(0) == dv_out.stride(0)
        check(lib().asr_attention_bwd_f32(_stream(), _p(q), _p(k), _p(v), _p(ctx), _p(d_ctx), _p(lse), _p(dq_out), dq_out.stride(0),
                                          _p(dk_out), _p(dv_out), dk_out.stride(0), B, h, Lq, Lk, _p(k_len), 1 if causal else 0,
                                          float(scale)), "asr_attention_bwd_f32")
        return
    assert q.dtype == torch.bfloat16 and ctx.is_contiguous() and d_ctx.is_contiguous() and d_ctx.dtype == torch.bfloat16
    assert dk_out.stride(0) == dv_out.stride(0) and dq_out.stride(1) == 1 and dk_out.stride(1) == 1
    delta = torch.empty((int(lib().asr_attention_bwd_workspace_floats(B, h, Lq)),), device=q.device, dtype=torch.float32)
    if drop_bits is None:
        drop_bits = attention_dropmask(drop, B, h, Lq, Lk, q.device)
    # two kernels, timed separately (algorithmic FLOPs on the 5-product count 10*B*h*64*Lq*Lk: dq owns dQ + one of the two
    # shared recomputed products, dkv owns dV, dK + the other)
    base = float(B) * h * 64 * Lq * Lk
    with _timed("attention_bwd_dq[B%d h%d %dx%d]" % (B, h, Lq, Lk), 4.0 * base):
        check(lib().asr_attention_bwd_dq(_stream(), _p(q), _p(k), _p(v), _p(ctx), _p(d_ctx), _p(lse), _p(delta), _p(dq_out),
                                         dq_out.stride(0), B, h, Lq, Lk, _p(k_len), 1 if causal else 0, float(scale), _d(drop),
                                         _p(drop_bits)), "asr_attention_bwd_dq")
    with _timed("attention_bwd_dkv[B%d h%d %dx%d]" % (B, h, Lq, Lk), 6.0 * base):
        check(lib().asr_attention_bwd_dkv(_stream(), _p(q), _p(k), _p(v), _p(d_ctx), _p(lse), _p(delta), _p(dk_out), _p(dv_out),
                                          dk_out.stride(0), B, h, Lq, Lk, _p(k_len), 1 if causal else 0, _d(drop), _p(drop_bits)),
              "asr_attention_bwd_dkv")


def embed_bwd(ids, dy, demb, drop=None):
    B, U = ids.shape if ids.dim() == 2 else (ids.shape[0], 1)
    D = dy.shape[-1]
    assert dy.numel() == B * U * D and dy.is_contiguous()
    check(lib().asr_embed_bwd(_stream(), _p(ids.contiguous()), _p(dy), B, U, D, demb.shape[0], _p(demb), _d(drop)), "asr_embed_bwd")


def adam_step(p, g, m, v, lr, beta1, beta2, eps, step, grad_scale=1.0, p16=None):
    _req_cuda(p, g, m, v, p16)
    assert p.is_contiguous() and g.is_contiguous() and p.dtype == torch.float32
    check(lib().asr_adam_step(_stream(), _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), float(lr), float(beta1), float(beta2),
                              float(eps), int(step), float(grad_scale)), "asr_adam_step")


def argmax_rows(x2d, out=None):
    """f32 [M, V] (row stride free) -> int64 [M], ties to the lowest index"""
    _req_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype == torch.float32
    M, V = x2d.shape
    if out is None:
        out = torch.empty(M, device=x2d.device, dtype=torch.int64)
    assert out.dtype == torch.int64 and out.numel() == M and out.is_contiguous()
    check(lib().asr_argmax_rows(_stream(), _p(x2d), x2d.stride(0), M, V, _p(out)), "asr_argmax_rows")
    return out


def topk_rows(x2d, k):
    """torch.topk(x, k, sorted=True) over the last dim of f32 [M, V] -> (values f32 [M, k], indices int64 [M, k]); ties in index order"""
    _req_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype == torch.float32
    M, V = x2d.shape
    vals = torch.empty((M, k), device=x2d.device, dtype=torch.float32)
    idx = torch.empty((M, k), device=x2d.device, dtype=torch.int64)
    check(lib().asr_topk_rows(_stream(), _p(x2d), x2d.stride(0), M, V, int(k), _p(vals), _p(idx)), "asr_topk_rows")
    return vals, idx


def lsm_topk_rows(x2d, k, twice=False):
    """topk_rows(log_softmax_rows(x), k) in one kernel; twice: of log_softmax applied twice (rows up to 4608 columns; longer rows
    chain the separate kernels)"""
    _req_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype == torch.float32
    M, V = x2d.shape
    if V > 4608:
        z = log_softmax_rows(x2d)
        return topk_rows(log_softmax_rows(z) if twice else z, k)
    vals = torch.empty((M, k), device=x2d.device, dtype=torch.float32)
    idx = torch.empty((M, k), device=x2d.device, dtype=torch.int64)
    check(lib().asr_lsm_topk_rows(_stream(), _p(x2d), x2d.stride(0), M, V, int(k), int(bool(twice)), _p(vals), _p(idx)), "asr_lsm_topk_rows")
    return vals, idx


def beam_prune(scores, next_scores, next_preds, beam):
    """decoder.py:196-209: scores f32 [B*beam], next_scores f32 / next_preds int64 [B*beam, beam] -> (new scores [B*beam],
    parent rows int64 [B*beam], new tokens int64 [B*beam])"""
    _req_cuda(scores, next_scores, next_preds)
    n = scores.numel()
    B = n // beam
    assert scores.is_contiguous() and next_scores.is_contiguous() and next_preds.is_contiguous() and next_scores.shape == (n, beam)
    ns = torch.empty(n, device=scores.device, dtype=torch.float32)
    parent = torch.empty(n, device=scores.device, dtype=torch.int64)
    tok = torch.empty(n, device=scores.device, dtype=torch.int64)
    check(lib().asr_beam_prune(_stream(), _p(scores), _p(next_scores), _p(next_preds), B, int(beam), _p(ns), _p(parent), _p(tok)),
          "asr_beam_prune")
    return ns, parent, tok


def decode_embed(cur, emb, pe, state, want_bf16=False):
    """x[b] = emb[cur[b]] + pe[state[0]] -> (y32 [B, D], y16 or None): the decoder input of the one new position, position on the device"""
    _req_cuda(cur, emb, pe, state)
    B, (V, D) = cur.numel(), emb.shape
    y32 = torch.empty((B, D), device=emb.device, dtype=torch.float32)
    y16 = torch.empty((B, D), device=emb.device, dtype=torch.bfloat16) if want_bf16 else None
    check(lib().asr_decode_embed(_stream(), _p(cur), _p(emb), _p(pe), _p(state), _p(y32), _p(y16), B, D, V, pe.shape[0]), "asr_decode_embed")
    return y32, y16


def kv_cache_put(k_new, v_new, k_cache, v_cache, state):
    """k_cache[:, :, state[0]] = k_new[:, :, 0] (same for v); caches [B, h, Tmax, 64], new heads [B, h, 1, 64] of the same dtype"""
    _req_cuda(k_new, v_new, k_cache, v_cache, state)
    B, h, Tmax, dk = k_cache.shape
    assert dk == 64 and k_new.is_contiguous() and v_new.is_contiguous() and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert k_new.dtype == k_cache.dtype == v_new.dtype == v_cache.dtype and k_new.numel() == B * h * 64
    check(lib().asr_kv_cache_put(_stream(), _p(k_new), _p(v_new), _p(k_cache), _p(v_cache), _p(state), B * h, Tmax, dtype_code(k_cache)),
          "asr_kv_cache_put")


def decode_advance(cur, preds, state, k_len, finished, len_decoded, eos_id):
    """decoder.py:151-158 on the device: append `cur` at preds[:, t + 1], update finished / len_decoded / k_len, t += 1"""
    _req_cuda(cur, preds, state, k_len, finished, len_decoded)
    B, Tp1 = preds.shape
    assert preds.is_contiguous() and preds.dtype == torch.int64 and finished.dtype == torch.uint8 and len_decoded.dtype == torch.int64
    assert k_len.dtype == torch.int32 and state.dtype == torch.int32
    check(lib().asr_decode_advance(_stream(), _p(cur), _p(preds), _p(state), _p(k_len), _p(finished), _p(len_decoded), int(eos_id), B, Tp1),
          "asr_decode_advance")


_DECODE_WS = {}


def decode_block_workspace(M, device):
    """the zeroed accumulator + arrival counters the fused decode sub-layers share (zeroed once; every launch leaves them zeroed)"""
    key = (torch.device(device).index, (M + 15) // 16, (M + 3) // 4)
    if key not in _DECODE_WS:
        _DECODE_WS[key] = torch.zeros(int(lib().asr_decode_block_workspace_bytes(int(M))), dtype=torch.uint8, device=device)
    return _DECODE_WS[key]


def decode_blocks_ok(x, d_ff=None, heads=None):
    """the fused decode sub-layers take d_model = 256, bf16 operands; the feed-forward one pays up to 64 rows (at 160 rows 31.6 us
    against 27.6 us for the four separate launches), the self-attention one up to 512"""
    return (x.f32.is_cuda and x.f32.shape[1] == 256 and x.f32.shape[0] <= (64 if d_ff is not None else 512) and x.f32.is_contiguous() and
            (d_ff is None or d_ff % 128 == 0) and (heads is None or 1 <= heads <= 16))


def decode_ffn(x16, x32, w1, b1, w2, b2, gamma, beta, eps, pre=None):
    """the feed-forward sub-layer of the decode step in one launch; pre = (Wo bf16 [256, 256], bo, gamma0, beta0, eps0): x16 is then the
    attention output and x32 the attention sub-layer's input - its output projection + residual + LayerNorm run in the same launch"""
    _req_cuda(x16, x32, w1, b1, w2, gamma, beta)
    M, D = x32.shape
    assert x16.dtype == torch.bfloat16 and w1.dtype == torch.bfloat16 and w2.dtype == torch.bfloat16 and x16.is_contiguous() and x32.is_contiguous()
    assert tuple(x16.shape) == (M, D)
    y32 = torch.empty((M, D), device=x32.device, dtype=torch.float32)
    y16 = torch.empty((M, D), device=x32.device, dtype=torch.bfloat16)
    ws = decode_block_workspace(M, x32.device)
    wo, bo, g0, bt0, eps0 = pre if pre is not None else (None, None, None, None, 0.0)
    if pre is not None:
        _req_cuda(wo, g0, bt0)
        assert wo.dtype == torch.bfloat16 and tuple(wo.shape) == (D, D)
    check(lib().asr_decode_ffn(_stream(), _p(x16), _p(x32), _p(w1), _p(b1), _p(w2), _p(b2), _p(gamma), _p(beta), _p(ws), _p(y32), _p(y16), M, D,
                               w1.shape[0], float(eps), _p(wo), _p(bo), _p(g0), _p(bt0), float(eps0)), "asr_decode_ffn")
    return y32, y16


def decode_self_attn(x16, x32, wqkv, bqkv, wo, bo, gamma, beta, k_cache, v_cache, state, eps, next_q=None):
    """the self-attention sub-layer of the decode step in one launch; next_q = (Wq bf16 [256, 256], bq, Lq, scale): also the next
    sub-layer's query projection of the output rows -> third result, bf16 head-major [M / Lq, 4, Lq, 64] (asr_proj_heads' layout)"""
    _req_cuda(x16, x32, wqkv, bqkv, wo, gamma, beta, k_cache, v_cache, state)
    M, D = x32.shape
    N, h, Tmax, dk = k_cache.shape
    assert N == M and dk == 64 and k_cache.dtype == torch.bfloat16 and v_cache.dtype == torch.bfloat16 and k_cache.is_contiguous() and v_cache.is_contiguous()
    assert x16.dtype == torch.bfloat16 and wqkv.dtype == torch.bfloat16 and wo.dtype == torch.bfloat16 and wqkv.shape == (3 * h * 64, D)
    y32 = torch.empty((M, D), device=x32.device, dtype=torch.float32)
    y16 = torch.empty((M, D), device=x32.device, dtype=torch.bfloat16)
    ws = decode_block_workspace(M, x32.device)
    wq2, bq2, q2, lq2, sc2 = None, None, None, 0, 0.0
    if next_q is not None:
        wq2, bq2, lq2, sc2 = next_q
        _req_cuda(wq2, bq2)
        assert wq2.dtype == torch.bfloat16 and tuple(wq2.shape) == (D, D) and M % lq2 == 0
        q2 = torch.empty((M // lq2, D // 64, lq2, 64), device=x32.device, dtype=torch.bfloat16)
    check(lib().asr_decode_self_attn(_stream(), _p(x16), _p(x32), _p(wqkv), _p(bqkv), _p(wo), _p(bo), _p(gamma), _p(beta), _p(k_cache), _p(v_cache),
                                     _p(state), _p(ws), _p(y32), _p(y16), M, D, h, Tmax, float(eps), _p(wq2), _p(bq2), _p(q2), int(lq2), float(sc2)),
          "asr_decode_self_attn")
    return (y32, y16, q2) if next_q is not None else (y32, y16)


def beam_cat_frames(frames, state, beam, other=None, cur=None, emb=None, pe=None):
    """rows [frames[b, t] | other[r]] or [frames[b, t] | emb[cur[r]] + pe[t]] (t = state[0], b = r // beam) -> f32 [N, D + D2]"""
    B, Tmax, D = frames.shape
    N = B * beam
    _req_cuda(frames, state)
    assert frames.is_contiguous() and frames.dtype == torch.float32
    if cur is not None:
        _req_cuda(cur, emb, pe)
        V, D2 = emb.shape
        out = torch.empty((N, D + D2), device=frames.device, dtype=torch.float32)
        check(lib().asr_beam_cat_frames(_stream(), _p(frames), _p(state), None, _p(cur), _p(emb), _p(pe), _p(out), N, beam, Tmax, D, D2, V,
                                        pe.shape[0]), "asr_beam_cat_frames")
    else:
        _req_cuda(other)
        assert other.is_contiguous() and other.dtype == torch.float32 and other.shape[0] == N
        D2 = other.shape[1]
        out = torch.empty((N, D + D2), device=frames.device, dtype=torch.float32)
        check(lib().asr_beam_cat_frames(_stream(), _p(frames), _p(state), _p(other), None, None, None, _p(out), N, beam, Tmax, D, D2, 0, 0),
              "asr_beam_cat_frames")
    return out


def beam_step(scores, next_scores, next_preds, preds, state, n_steps, parent, cur, beam):
    """one pruning step of every live utterance, in place (scores [N], preds int64 [N, W], cur, parent int64 [N])"""
    _req_cuda(scores, next_scores, next_preds, preds, state, n_steps, parent, cur)
    N, W = preds.shape
    assert preds.is_contiguous() and preds.dtype == torch.int64 and next_preds.dtype == torch.int64 and next_scores.shape == (N, beam)
    assert n_steps.dtype == torch.int32 and state.dtype == torch.int32 and n_steps.numel() * beam == N and scores.dtype == torch.float32
    check(lib().asr_beam_step(_stream(), _p(scores), _p(next_scores), _p(next_preds), _p(preds), _p(state), _p(n_steps), _p(parent), _p(cur),
                              N // beam, int(beam), W), "asr_beam_step")


def beam_reorder_cache(cache, parent, state, beam):
    """cache [n_kv, N, h, Tmax, 64] re-gathered along N by `parent`, in place, positions <= state[0]"""
    _req_cuda(cache, parent, state)
    n_kv, N, h, Tmax, dk = cache.shape
    assert dk == 64 and cache.is_contiguous() and parent.dtype == torch.int64
    check(lib().asr_beam_reorder_cache(_stream(), _p(cache), _p(parent), _p(state), n_kv, N // beam, int(beam), h, Tmax, dtype_code(cache)),
          "asr_beam_reorder_cache")


def beam_advance(state, k_len, cur=None, eos_id=0, finished=None, len_decoded=None):
    """end of a search step (state[1] must be -1 while the search runs); with `finished` also the <eos> bookkeeping of decoder.py:211-216"""
    _req_cuda(state, k_len)
    assert state.dtype == torch.int32 and k_len.dtype == torch.int32
    if finished is not None:
        _req_cuda(cur, finished, len_decoded)
        assert finished.dtype == torch.uint8 and len_decoded.dtype == torch.int64 and cur.dtype == torch.int64
    check(lib().asr_beam_advance(_stream(), _p(state), _p(k_len), k_len.numel(), _p(cur), int(eos_id), _p(finished), _p(len_decoded)),
          "asr_beam_advance")


def log_softmax_rows(x2d):
    _req_cuda(x2d)
    assert x2d.dim() == 2 and x2d.stride(1) == 1 and x2d.dtype == torch.float32
    M, V = x2d.shape
    y = torch.empty((M, V), device=x2d.device, dtype=torch.float32)
    check(lib().asr_log_softmax_rows(_stream(), _p(x2d), x2d.stride(0), M, V, _p(y), V), "asr_log_softmax_rows")
    return y


def ctc_greedy_reduce(frames, lens, blank):
    """frames int64 [B, L] (per-frame argmax), lens int [B] -> (tokens int64 [B, L] zero-padded, n_tokens int32 [B])"""
    _req_cuda(frames, lens)
    frames = frames.to(torch.int64).contiguous()
    B, L = frames.shape
    out = torch.empty((B, L), device=frames.device, dtype=torch.int64)
    n = torch.empty(B, device=frames.device, dtype=torch.int32)
    check(lib().asr_ctc_greedy_reduce(_stream(), _p(frames), _p(as_i32(lens, frames.device)), B, L, int(blank), _p(out), _p(n)),
          "asr_ctc_greedy_reduce")
    return out, n


def add_(dst, src):
    """dst += src for f32 tensors viewed as [rows, cols] (last dim contiguous, rows uniformly strided) - on the HIP path."""
    _req_cuda(dst, src)
    assert dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.shape == src.shape
    cols = dst.shape[-1] if dst.dim() else 1

    def rows_of(t):       # -> (row stride, rows) of t seen as rows of `cols` contiguous elements
        if t.is_contiguous():
            return cols, t.numel() // cols
        assert t.dim() == 2 and t.stride(1) == 1, "add_: rows must be contiguous"
        return t.stride(0), t.shape[0]
    (ldd, rows), (lds, rows_s) = rows_of(dst), rows_of(src)
    assert rows == rows_s
    check(lib().asr_add2d(_stream(), _p(dst), ldd, _p(src), lds, rows, cols), "asr_add2d")
    return dst


def add_transposed_(dst, src, O, A, Bn, lds=None):
    """dst (contiguous, viewed [O, A, Bn]) += src[o, b, a] (src rows of `lds` >= A*Bn elements)."""
    _req_cuda(dst, src)
    assert dst.is_contiguous() and dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.numel() == O * A * Bn
    lds = A * Bn if lds is None else lds
    check(lib().asr_add_transposed(_stream(), _p(dst), _p(src), O, A, Bn, lds), "asr_add_transposed")
    return dst


def relu_mask_mul(d, y, out=None):
    """d * (y > 0) (f32 d, f32 / bf16 y, same numel, both contiguous)"""
    _req_cuda(d, y)
    assert d.is_contiguous() and y.is_contiguous() and d.numel() == y.numel() and d.dtype == torch.float32
    out = torch.empty_like(d) if out is None else out
    check(lib().asr_relu_mask_mul(_stream(), _p(d), _p(y), dtype_code(y), _p(out), d.numel()), "asr_relu_mask_mul")
    return out


def conv1d_overlap_add(d_win, rows, w, cin):
    _req_cuda(d_win)
    assert d_win.is_contiguous() and d_win.dtype == torch.float32 and d_win.shape == (rows, w * cin)
    d_in = torch.empty((rows + w, cin), device=d_win.device, dtype=torch.float32)
    check(lib().asr_conv1d_overlap_add(_stream(), _p(d_win), rows, w, cin, _p(d_in)), "asr_conv1d_overlap_add")
    return d_in


def assigner_tail_bwd(g, alpha, h, w, B, L, dw, db):
    """-> d_h [B*L, Dh]; dw [Dh] / db [1] accumulated in place."""
    _req_cuda(g, alpha, h, w, dw, db)
    Dh = h.shape[-1]
    g, alpha, h, w = g.contiguous().float(), alpha.contiguous(), h.contiguous(), w.contiguous()
    assert dw.is_contiguous() and dw.numel() == Dh and db.numel() == 1 and h.numel() == B * L * Dh
    d_h = torch.empty((B * L, Dh), device=h.device, dtype=torch.float32)
    check(lib().asr_assigner_tail_bwd(_stream(), _p(g), _p(alpha), _p(h), _p(w), B, L, Dh, _p(d_h), _p(dw), _p(db)), "asr_assigner_tail_bwd")
    return d_h


def cif_rescale_fwd(alpha_raw, targets, noise):
    """-> (alpha, num_pred, num, scale): cif_model.py:44-48"""
    _req_cuda(alpha_raw, targets, noise)
    B, L = alpha_raw.shape
    alpha_raw, targets, noise = alpha_raw.contiguous(), targets.to(torch.int64).contiguous(), noise.float().contiguous()
    alpha = torch.empty_like(alpha_raw)
    num_pred, num, scale = (torch.empty(B, device=alpha_raw.device, dtype=torch.float32) for _ in range(3))
    check(lib().asr_cif_rescale_fwd(_stream(), _p(alpha_raw), _p(targets), _p(noise), B, L, targets.shape[1], _p(alpha), _p(num_pred),
                                    _p(num), _p(scale)), "asr_cif_rescale_fwd")
    return alpha, num_pred, num, scale


def cif_rescale_bwd(d_alpha, alpha_raw, scale, num_pred, d_num_in=None):
    _req_cuda(d_alpha, alpha_raw, scale, num_pred, d_num_in)
    B, L = alpha_raw.shape
    d_alpha = d_alpha.contiguous()
    dn = d_num_in.contiguous().float() if d_num_in is not None else None
    d_raw = torch.empty_like(alpha_raw)
    check(lib().asr_cif_rescale_bwd(_stream(), _p(d_alpha), _p(alpha_raw), _p(scale), _p(num_pred), _p(dn), B, L, _p(d_raw)),
          "asr_cif_rescale_bwd")
    return d_raw


def step_tick(state, k, init_lr, warmup, beta1, beta2):
    """state: int32 [8] device tensor (asr_hip.h: asr_step_tick) - step += 1, Noam lr and Adam bias corrections recomputed on device."""
    _req_cuda(state)
    check(lib().asr_step_tick(_stream(), _p(state), float(k), float(init_lr), float(warmup), float(beta1), float(beta2)), "asr_step_tick")


def adam_step_dev(p, g, m, v, state, beta1, beta2, eps, grad_scale=1.0, p16=None):
    """adam_step with (lr, bias corrections) read from the device step state (hipGraph replay)."""
    _req_cuda(p, g, m, v, p16, state)
    check(lib().asr_adam_step_dev(_stream(), _p(p), _p(g), _p(m), _p(v), _p(p16), p.numel(), _p(state), float(beta1), float(beta2),
                                  float(eps), float(grad_scale)), "asr_adam_step_dev")


def conv_im2col(x, C, Tout, Fout, ldc, out_dtype):
    """x channel-last [B,Tin,Fin,C] (f32 or bf16) -> patch matrix [B*Tout*Fout, ldc] (columns 9*C.. zero)."""
    _req_cuda(x)
    B, Tin, Fin = x.shape[0], x.shape[1], x.shape[2]
    col = torch.empty((B * Tout * Fout, ldc), device=x.device, dtype=out_dtype)
    check(lib().asr_conv_im2col(_stream(), _p(x), dtype_code(x), C, _p(col), dtype_code(col), ldc, B, Tin, Fin, Tout, Fout), "asr_conv_im2col")
    return col


def conv_col2im_relu(dcol, y, Tout, Fout):
    """dcol bf16 [B*Tout*Fout, 288(+pad)], y bf16 [B,Tin,Fin,32] (forward output, ReLU mask) -> dx bf16 [B,Tin,Fin,32]."""
    B, Tin, Fin, _ = y.shape
    dx = torch.empty_like(y)
    assert dcol.dtype == y.dtype and y.is_contiguous()
    if y.dtype == torch.float32:       # fp32 parity mode
        check(lib().asr_conv_col2im_relu_f32(_stream(), _p(dcol), dcol.stride(0), _p(y), _p(dx), B, Tin, Fin, Tout, Fout),
              "asr_conv_col2im_relu_f32")
        return dx
    check(lib().asr_conv_col2im_relu(_stream(), _p(dcol), dcol.stride(0), _p(y), _p(dx), B, Tin, Fin, Tout, Fout), "asr_conv_col2im_relu")
    return dx


def conv_sub1_bwd_x(dy, w, xin, Tout, Fout):
    """dy bf16 [B*Tout*Fout, 32] (any shape with that layout), w f32 [32,32,3,3], xin bf16 [B,Tin,Fin,32] -> dx bf16 like xin (ReLU-masked)"""
    _req_cuda(dy, w, xin)
    B, Tin, Fin, _ = xin.shape
    assert dy.dtype == torch.bfloat16 and xin.dtype == torch.bfloat16 and dy.is_contiguous() and xin.is_contiguous()
    assert dy.numel() == B * Tout * Fout * 32 and w.dtype == torch.float32 and w.is_contiguous() and tuple(w.shape) == (32, 32, 3, 3)
    dx = torch.empty_like(xin)
    check(lib().asr_conv_sub1_bwd_x(_stream(), _p(dy), _p(w), _p(xin), _p(dx), B, Tin, Fin, Tout, Fout), "asr_conv_sub1_bwd_x")
    return dx


def conv_sub1_bwd_w(dy, x, Tout, Fout, db=None):
    """-> dw f32 [32, 288] (column tap*32 + ci) of dy bf16 [B,Tout,Fout,32] against the layer input x bf16 [B,Tin,Fin,32]; db (f32 [32]) += sum dy"""
    _req_cuda(dy, x, db)
    B, Tin, Fin, _ = x.shape
    assert dy.dtype == torch.bfloat16 and x.dtype == torch.bfloat16 and dy.is_contiguous() and x.is_contiguous()
    assert dy.numel() == B * Tout * Fout * 32
    dw = torch.zeros((32, 288), device=x.device, dtype=torch.float32)
    ws = torch.empty(int(lib().asr_conv_sub1_bwd_w_workspace_floats()), device=x.device, dtype=torch.float32)
    check(lib().asr_conv_sub1_bwd_w(_stream(), _p(dy), _p(x), _p(dw), _p(db), _p(ws), B, Tin, Fin, Tout, Fout), "asr_conv_sub1_bwd_w")
    return dw


def conv_sub0_bwd_w(dy, feats, dw, db, T1, F1):
    """dw f32 [32,1,3,3] += and db f32 [32] += of the first conv layer from dy bf16 [B,T1,F1,32] and the features f32 [B,T,D]"""
    _req_cuda(dy, feats, dw, db)
    B, T, D = feats.shape
    assert dy.dtype == torch.bfloat16 and dy.is_contiguous() and dy.numel() == B * T1 * F1 * 32
    assert feats.dtype == torch.float32 and feats.is_contiguous() and dw.is_contiguous() and dw.numel() == 288 and dw.dtype == torch.float32
    ws = torch.empty(int(lib().asr_conv_sub1_bwd_w_workspace_floats()), device=feats.device, dtype=torch.float32)
    check(lib().asr_conv_sub0_bwd_w(_stream(), _p(dy), _p(feats), _p(dw), _p(db), _p(ws), B, T, D, T1, F1), "asr_conv_sub0_bwd_w")


def cif_bwd(hidden, cur, rem, tok, n_fire, d_out):
    """-> (d_hidden [B,L,H], d_alpha [B,L]) for out = cif(hidden, alpha)"""
    hidden = hidden.contiguous()
    d_out = d_out.contiguous().float()
    B, L, H = hidden.shape
    Umax = d_out.shape[1]
    d_hidden = torch.empty_like(hidden)
    d_cur = torch.empty((B, L), device=hidden.device, dtype=torch.float32)
    d_rem = torch.empty((B, L), device=hidden.device, dtype=torch.float32)
    check(lib().asr_cif_gather_bwd(_stream(), _p(hidden), _p(cur), _p(rem), _p(tok), _p(n_fire), _p(d_out), B, L, H, Umax, _p(d_hidden),
                                   _p(d_cur), _p(d_rem)), "asr_cif_gather_bwd")
    d_alpha = torch.empty((B, L), device=hidden.device, dtype=torch.float32)
    check(lib().asr_cif_scan_bwd(_stream(), _p(d_cur), _p(d_rem), _p(tok), B, L, _p(d_alpha)), "asr_cif_scan_bwd")
    return d_hidden, d_alpha


# ---- gradient all-reduce from the C launch loop (asr_hip.h: asr_rccl_*, asr_collective_mark; csrc/collective.hip) ----------------------
class _DefaultGroupKey:
    pass


_DEFAULT_GROUP_KEY = _DefaultGroupKey()
_RCCL = {"loaded": False, "comms": weakref.WeakKeyDictionary()}


def _rccl_teardown():
    """atexit: ncclCommDestroy for every communicator the library created (before the runtime goes away)"""
    try:
        for c in list(_RCCL["comms"].values()):
            c.destroy()
    except Exception:
        pass


atexit.register(_rccl_teardown)
_COLLECTIVE_CB = ctypes.CFUNCTYPE(ctypes.c_int, ctypes.c_void_p, ctypes.c_void_p, ctypes.c_longlong, ctypes.c_int, ctypes.c_void_p)


def rccl_load():
    """Bind RCCL's C API inside libasr_hip.so: the librccl torch has already mapped (one RCCL per process), else ROCm's."""
    if not _RCCL["loaded"]:
        # one RCCL per process: whatever librccl is ALREADY mapped (torch's own copy, under whatever file name) is the one to bind -
        # by its mapped path, so that the loader cannot bring in a second copy under another name
        mapped = None
        try:
            with open("/proc/self/maps") as f:
                for line in f:
                    if "librccl" in line and "/" in line:
                        mapped = line[line.index("/"):].strip()
                        break
        except OSError:
            pass
        cands = [mapped, os.path.join(os.path.dirname(torch.__file__), "lib", "librccl.so"), "/opt/rocm/lib/librccl.so.1"]
        path = next((c for c in cands if c and os.path.exists(c)), None)
        check(lib().asr_rccl_load(path.encode() if path else None), "asr_rccl_load")
        _RCCL["loaded"] = True


ERR_COLLECTIVE_STEP = -6      # asr_hip.h: ASR_ERR_COLLECTIVE_STEP


class RcclComm:
    """An RCCL communicator owned by libasr_hip.so (not torch's): what the graph executor's collective nodes all-reduce through."""

    def __init__(self, handle, world, rank):
        self.handle, self.world, self.rank = handle, world, rank

    def all_reduce_(self, t):
        """t (f32, contiguous, cuda) <- sum over ranks, queued on the current stream"""
        _req_cuda(t)
        if t.dtype != torch.float32 or not t.is_contiguous():
            raise RuntimeError("RcclComm.all_reduce_: a contiguous float32 tensor is required")
        self._live()
        check(lib().asr_rccl_all_reduce_f32(self.handle, ctypes.c_void_p(t.data_ptr()), t.numel(), _stream()), "asr_rccl_all_reduce_f32")
        return t

    def check(self):
        self._live()
        check(lib().asr_rccl_comm_check(self.handle), "asr_rccl_comm_check")

    def destroy(self):
        h, self.handle = self.handle, None
        if h:
            lib().asr_rccl_comm_destroy(h)

    def abort(self):
        """asr_rccl_comm_abort: tear down without waiting (peers get an asynchronous error); the handle is dropped and the object evicted
        from the per-group cache, so neither the atexit teardown nor a later rccl_comm() touches the freed communicator."""
        h, self.handle = self.handle, None
        for k in [k for k, c in list(_RCCL["comms"].items()) if c is self]:
            del _RCCL["comms"][k]
        if h:
            lib().asr_rccl_comm_abort(h)

    def _live(self):
        if not self.handle:
            raise RuntimeError("RcclComm: this communicator was aborted or destroyed")


def rccl_comm(group=None, device=None):
    """The RcclComm over the ranks of `group` (default group when None; a 1-rank communicator when torch.distributed is not
    initialised).  Collective over the group on first use: rank 0's unique id travels through torch.distributed (bootstrap only)."""
    import torch.distributed as dist
    # keyed on the group OBJECT (weakly): id() of a destroyed group can be handed to a new one, which would then get a communicator
    # over the old group's ranks
    key = group if group is not None else _DEFAULT_GROUP_KEY
    if key in _RCCL["comms"]:
        return _RCCL["comms"][key]
    rccl_load()
    world, rank = 1, 0
    if dist.is_available() and dist.is_initialized():
        world, rank = dist.get_world_size(group), dist.get_rank(group)
    uid = (ctypes.c_char * 128)()
    if rank == 0:
        check(lib().asr_rccl_unique_id(uid), "asr_rccl_unique_id")
    if world > 1:
        box = [bytes(uid.raw)]
        src = dist.get_global_rank(group, 0) if group is not None else 0
        dist.broadcast_object_list(box, src=src, group=group)
        uid = (ctypes.c_char * 128).from_buffer_copy(box[0])
    h = ctypes.c_void_p()
    with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
        check(lib().asr_rccl_comm_create(uid, world, rank, ctypes.byref(h)), "asr_rccl_comm_create")
    comm = RcclComm(h, world, rank)
    if world > 1:      # pre-flight: the sum of (rank + 1) over the ranks, through the communicator the step will use
        with torch.cuda.device(device if device is not None else torch.cuda.current_device()):
            probe = torch.full((256,), float(rank + 1), device="cuda")
            comm.all_reduce_(probe)
            got = float(probe[0])
        if got != world * (world + 1) / 2:
            raise RuntimeError("asr_amd: RCCL pre-flight all-reduce returned %r over %d ranks" % (got, world))
    _RCCL["comms"][key] = comm
    return comm


def collective_mark(t, tag=0):
    """Bucket-ready marker for t (a contiguous f32 view) on the current stream: a no-op kernel that a CAPTURED step's executor turns
    into the all-reduce of t (asr_collective_mark)."""
    _req_cuda(t)
    if t.dtype != torch.float32 or not t.is_contiguous():
        raise RuntimeError("collective_mark: a contiguous float32 tensor is required")
    check(lib().asr_collective_mark(ctypes.c_void_p(t.data_ptr()), t.numel(), int(tag), _stream()), "asr_collective_mark")


def torch_collective_fn(tensors, group):
    """An asr_collective_fn that all-reduces through torch.distributed (the test rig: gloo ranks sharing one GPU, which RCCL does not
    accept as peers).  tensors: the f32 buffers the marked views live in.  The host blocks in each call - a rig, not the product path."""
    import torch.distributed as dist
    state = {"error": None}

    def fn(ctx, buf, count, tag, stream):
        try:
            for t in tensors:
                off = (buf - t.data_ptr()) // 4
                if 0 <= off and off + count <= t.numel() and (buf - t.data_ptr()) % 4 == 0:
                    # (ctypes hands the null stream over as None.  It is torch's default stream and must be named as that: work queued
                    # through ExternalStream(0) was seen to overtake kernels the C loop had launched on the null stream before it,
                    # once a hipGraphLaunch had run on that stream - tools/probe/external_stream_order.py)
                    s = torch.cuda.ExternalStream(stream, device=t.device) if stream else torch.cuda.default_stream(t.device)
                    with torch.cuda.stream(s):
                        dist.all_reduce(t.view(-1)[off:off + count], group=group)
                    return 0
            raise RuntimeError("collective node buffer %#x (+%d floats) is in none of the registered tensors" % (buf, count))
        except Exception as e:      # (an exception cannot cross the C loop)
            state["error"] = e
            return -1
    cb = _COLLECTIVE_CB(fn)
    cb.state = state
    return cb


class GraphExec:
    """Multi-stream executor over a captured HIP graph (asr_hip.h: asr_graphx_*).  Keeps the torch graph (the hipGraph_t, the nodes'
    parameter blocks and the capture's memory pool live in it) and launches its nodes itself on a few free-running streams."""

    def __init__(self, handle, graph, info):
        self._h, self._graph, self.info = handle, graph, info

    @classmethod
    def from_torch_graph(cls, graph, max_streams=4):
        """-> GraphExec, or None (with a warning that says why) when the graph holds nodes the executor does not launch.
        max_streams = 4: one per hardware queue of the runtime - with 8, two of the executor's streams shared each queue and three of the
        four rotations of the stream map put the weight-gradient chain on the launch stream's queue (S1 13.6 vs 12.6 ms, S2 9-10 vs 6.9);
        with 4 every rotation is good (12.35-12.45, 6.7)."""
        raw = graph.raw_cuda_graph()
        h = ctypes.c_void_p()
        rc = lib().asr_graphx_create(ctypes.c_void_p(int(raw)), int(max_streams), ctypes.byref(h))
        if rc != 0:
            import warnings
            warnings.warn("asr_amd: the multi-stream graph executor does not take this graph (%s); replaying it with hipGraphLaunch" %
                          lib().asr_last_error().decode(errors="replace"))
            return None
        n = [ctypes.c_int() for _ in range(4)]
        lib().asr_graphx_info(h, *[ctypes.byref(v) for v in n])
        nc, tot = ctypes.c_int(), ctypes.c_longlong()
        lib().asr_graphx_collectives(h, ctypes.byref(nc), ctypes.byref(tot))
        return cls(h, graph, dict(nodes=n[0].value, kernels=n[1].value, streams=n[2].value, events=n[3].value, collectives=nc.value,
                                  collective_floats=tot.value))

    def set_collective(self, comm=None, fn=None):
        """What the plan's collective nodes call: an RcclComm, or an asr_collective_fn callback (torch_collective_fn)."""
        self._comm, self._fn = comm, fn      # (kept alive with the executor)
        if comm is not None:
            comm._live()
        check(lib().asr_graphx_set_collective(self._h, comm.handle if comm is not None else None,
                                              ctypes.cast(fn, ctypes.c_void_p) if fn is not None else None, None), "asr_graphx_set_collective")

    def launch(self):
        rc = lib().asr_graphx_launch(self._h, _stream())
        fn = getattr(self, "_fn", None)
        if rc != 0 and fn is not None and fn.state["error"] is not None:
            err, fn.state["error"] = fn.state["error"], None
            raise err
        if rc == ERR_COLLECTIVE_STEP:
            # the executor has forgotten the communicator it borrowed; its owner (this side) aborts it and drops every reference
            msg = lib().asr_last_error().decode(errors="replace")
            comm, self._comm = getattr(self, "_comm", None), None
            if comm is not None:
                comm.abort()
            raise RuntimeError("asr_graphx_launch failed (rc=%d): %s" % (rc, msg))
        check(rc, "asr_graphx_launch")

    def place_streams(self, clear=False):
        """Explicit logical -> physical side-stream map from hardware-queue probes (asr_graphx_place_streams); clear: back to the rotation."""
        check(lib().asr_graphx_place_streams(self._h, _stream(), 1 if clear else 0), "asr_graphx_place_streams")

    def set_rotation(self, r):
        """Logical side stream i -> physical side stream (i + r) mod n: streams share hardware queues, which ones is not ours to choose."""
        check(lib().asr_graphx_set_rotation(self._h, int(r)), "asr_graphx_set_rotation")
        self.rotation = int(r)

    def __del__(self):
        h, self._h = getattr(self, "_h", None), None
        if h:
            try:      # (at interpreter shutdown the module's globals may be gone already: nothing left to release then)
                torch.cuda.synchronize()
                lib().asr_graphx_destroy(h)
            except Exception:
                pass
