"""nn.Module mirrors of the reference's model classes, running on libasr_hip.so.

Constructor signatures, forward signatures / return tuples and state_dict keys follow the reference
(each class cites its reference file:line), so a reference checkpoint's `state_dict`
loads unchanged.  What differs is the execution: masks are never materialised (kernels take lengths),
Q/K/V projections are one head-major GEMM, attention is a flash-style MFMA kernel, residual+LayerNorm+
length-mask is one row kernel, and the attention-probability tensor the reference returns (and every
caller discards) is not produced (`None` is returned in its place).

Precision: "bf16" (default; bf16 MFMA operands, fp32 accumulate / softmax / LayerNorm / losses) or "f32"
(exact-fp32 MFMA, the parity mode).  Select with `set_precision()` or the `precision(...)` context.
Dropout (training mode, rate > 0): fused into the kernels at the reference's seven sites (attention probabilities, fc / w_2
outputs before the residual add, encoder input, decoder embedding, assigner) as a counter-based hash mask that the backward
kernels regenerate - see `asr_dropout_t` in include/asr_hip.h.  Keys derive from (seed, qualified module name, call count):
`manual_seed()` makes runs reproducible; the mask stream is not torch's (no implementation could match that bit for bit).
"""
import contextlib
import math
import os
import zlib

import torch
import torch.nn as nn

from . import ops

_PRECISION = os.environ.get("ASR_AMD_PRECISION", "bf16")
_LOG2E = 1.4426950408889634
# 1 = the attention sub-layer's LayerNorm backward inside the feed-forward sub-layer's data-gradient launch (asr_ffn_bwd_ln).  Off: measured
# neutral at S1 (11.19 / 11.29 against 11.12 / 11.28 ms) - the launch grows by 24.5 us (129.4 against 104.9) where it saves a 19 us kernel
# and a 5 us gap: ffn_bwd is ONE round of 250 workgroups at one wave per SIMD, so the 82 MB the epilogue still moves (s in, ds and
# ds16 out) go through a chip that is doing nothing else, while the stand-alone kernel runs beside the weight-gradient stream.
_FOLD_LN = False
# 1 = the one-launch sub-layers at encoder size (asr_ffn_fwd, asr_proj_ln_fwd) do not store the pre-norm sum: the LayerNorm's backward takes
# x^ = (y - beta) / gamma from the OUTPUT, which stays alive anyway (33 MB less to write and to hold per LayerNorm, 24 of them per S1 step:
# 0.8 GB of activations).  Off by default: a memory option, not a faster step (11.27-11.50 against 11.22-11.26 ms on one box).
_LN_FROM_Y = os.environ.get("ASR_AMD_LN_FROM_Y", "0") == "1"
_FOLD_LN_QKV = True      # the feed-forward sub-layer's LayerNorm backward as the epilogue of the next layer's Q/K/V data gradient (asr_dgrad_rows_ln)
_MASK_PREFETCH = True      # (bench.py's kernel-alone timing pass sets this False: the masks are then hashed in line)
_MASK_GROUP = True      # short sequences: the encoder's attention-dropout masks 8 sites per launch


def set_precision(p):
    global _PRECISION
    assert p in ("bf16", "f32")
    _PRECISION = p
    ops.EXACT_F32 = p == "f32"


def get_precision():
    return _PRECISION


@contextlib.contextmanager
def precision(p):
    old = get_precision()
    set_precision(p)
    try:
        yield
    finally:
        set_precision(old)


def _cdtype():
    return torch.bfloat16 if _PRECISION == "bf16" else torch.float32


# ---- dropout keys -----------------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1
_DROP_STATE = {"seed": None, "calls": {}, "salt": None}     # salt: device pointer of the step counter while a step is captured


def manual_seed(seed):
    """Seed of the dropout masks (and reset of the per-site call counters): two runs with the same seed and the same sequence
    of forward calls drop the same elements."""
    _DROP_STATE["seed"] = int(seed) & _M64
    _DROP_STATE["calls"] = {}


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def dropout_site_keys(seed, name, call):
    """(key0, key1) of asr_dropout_t for dropout site `name` (the reference's qualified nn.Dropout module name, e.g.
    "encoder.layer_stack.0.slf_attn.attention.dropout") on its `call`-th invocation (1-based)."""
    h = _splitmix64(seed & _M64)
    h = _splitmix64(h ^ zlib.crc32(name.encode()))
    h = _splitmix64(h ^ (call & _M64))
    return h & 0xFFFFFFFF, h >> 32


def dropout_thr16(p):
    return int(round(float(p) * 65536.0))


def _assign_names(root):
    """Qualified names for the dropout keys (done by the top-level model's forward; a module used stand-alone has the empty prefix)."""
    if root.__dict__.get("_asr_named"):
        return
    for name, m in root.named_modules():
        m.__dict__["_asr_name"] = name
    root.__dict__["_asr_named"] = True


def _drop(mod, suffix):
    """asr_dropout_t of dropout site `<module name>.<suffix>` for this call, or None when inactive (eval mode / rate 0)."""
    p = float(getattr(mod, "dropout_rate", 0.0) or 0.0)
    if not mod.training or p <= 0.0:
        return None
    if _PRECISION != "bf16":
        raise RuntimeError("dropout (training mode) runs on the bf16 path; the f32 parity mode needs .eval() or dropout = 0")
    thr = dropout_thr16(p)
    if thr <= 0:
        return None
    if thr >= 65536:
        raise ValueError("dropout rate must be < 1")
    if _DROP_STATE["seed"] is None:
        manual_seed(torch.initial_seed())
    prefix = mod.__dict__.get("_asr_name", "")
    name = (prefix + "." + suffix) if prefix else suffix
    calls = _DROP_STATE["calls"]
    key = (id(mod), suffix)
    calls[key] = calls.get(key, 0) + 1
    k0, k1 = dropout_site_keys(_DROP_STATE["seed"], name, calls[key])
    return ops.Dropout(thr, k0, k1, _DROP_STATE["salt"])


# Workgroups of a weight-gradient launch on the trainer's side stream: half the CUs.  A slab weight-gradient workgroup takes a whole CU
# (512 registers per lane, 128 KiB of LDS), so with 256 of them the main chain's next kernel waits for CUs to come free; with 128 it
# always finds half the chip (S1 replay 11.98-12.05 -> 11.84-11.91 ms on two boxes; 96: 11.92, 64: 12.16, 192: 11.96).
_WGRAD_SIDE_WGS = 128
_WGRAD = None      # set by Trainer.backward: {"stream": side stream, "keep": [operands kept alive until the streams join]}


def flush_wgrads():
    """Issues the decoder-sized weight gradients collected by `_wg` as one grouped launch on the side stream.  Called when 8 are
    pending, before a gradient bucket's all-reduce is queued, and at the end of the backward."""
    if _WGRAD is None or not _WGRAD.get("pending"):
        return
    pend, _WGRAD["pending"] = _WGRAD["pending"], []
    main, side = torch.cuda.current_stream(), _WGRAD["stream"]
    ops.order_after(side, main)
    with torch.cuda.stream(side):
        if len(pend) == 1:
            a, b, out, acc, cs = pend[0]
            ops.gemm_tn(a, b, out=out, accumulate=acc, colsum=cs, max_wgs=_WGRAD_SIDE_WGS)
        else:
            ops.gemm_tn_group(pend)


def _wg(a, b, **kw):
    """Weight-gradient GEMM of an encoder / decoder layer.  Nothing later in the backward reads its result, so under the trainer
    it is queued on a side stream behind an event of the main one (the operands exist by then) and overlaps the main chain.  The
    decoder's (1632 rows: a launch pair each, 13-17 us of which ~2 are work) are collected and issued eight at a time."""
    if _WGRAD is None:
        return ops.gemm_tn(a, b, **kw)
    out = kw.get("out")
    if ops.gemm_tn_group_ok(a, b, out) and set(kw) <= {"out", "accumulate", "colsum"}:
        _WGRAD["keep"].append((a, b))
        if any(p[2].data_ptr() == out.data_ptr() for p in _WGRAD.get("pending", ())):
            flush_wgrads()        # the same destination twice (tied weights): two problems of one launch would race on it and its workspace
        _WGRAD.setdefault("pending", []).append((a, b, out, bool(kw.get("accumulate", False)), kw.get("colsum")))
        if len(_WGRAD["pending"]) >= 8:
            flush_wgrads()
        return out
    # (Round 5, measured and not kept: ALL of an encoder layer's weight gradients - or two layers' - as one grouped launch with a
    # workgroup budget of its own, ops.gemm_tn_group(group_wgs=).  Alone on the chip the batch wins - 133 us per layer at 256 workgroups
    # against 173 us for the four launch pairs, tools/bench_wgrad_batch.py - but in the step every variant lost 0.1-0.25 ms (11.20-11.36
    # against 11.06-11.13 ms): one long launch holds its CUs against the main chain longer than four short ones do.  Unsplit over M
    # - one workgroup per output tile, no slabs, no reduce launch - is slower still, 317 us per layer: a workgroup that walks all
    # 32 000 rows alone misses L2 on every operand tile and its three-step prefetch ring does not cover an HBM round trip.)
    main, side = torch.cuda.current_stream(), _WGRAD["stream"]
    ops.order_after(side, main)
    _WGRAD["keep"].append((a, b))
    with torch.cuda.stream(side):
        return ops.gemm_tn(a, b, max_wgs=_WGRAD_SIDE_WGS, **kw)


def _prefetch_attn_masks(sites, training, device):
    """sites: [(MultiheadAttention, B, Lq, Lk)] in call order -> per site None or (asr_dropout_t, keep bits, event) for
    MultiheadAttention._impl(attn_drop=).  The attention-dropout keep bits of every listed call are hashed on a side stream while
    the main stream runs the layers before them: the masks depend on (seed, site, call count, shape) only - not on data - and the
    hash is pure integer VALU work (~50-60 us per [32, 4, 1000, 1000] layer alone on the chip), which co-runs with the MFMA /
    HBM-bound kernels of the main stream."""
    none = [None] * len(sites)
    if not (_MASK_PREFETCH and training and device.type == "cuda"):
        return none
    drops = [_drop(m, "attention.dropout") for m, _, _, _ in sites]
    if all(d is None for d in drops):
        return none
    main, aux = torch.cuda.current_stream(), ops.aux_stream(device, slot=1)
    aux.wait_stream(main)      # fork from the main stream (ordering after the previous step's readers; required under graph capture)
    out = []
    shapes = {(B, m.n_head, Lq, Lk) for m, B, Lq, Lk in sites}
    (B0, h0, Lq0, Lk0) = next(iter(shapes))
    # short sequences (L = T/4 behind the conv front end: ~5-10 us of hashing per site): up to 8 sites per launch and ONE event per
    # group - a step of 6-7 ms is queued by the host about as fast as the GPU runs it, and 12 launches + 12 event pairs at its very
    # start left the first two encoder layers waiting for the host (0.5 ms of idle GPU per CIF_Model step in tools/timeline.py)
    if _MASK_GROUP and len(shapes) == 1 and all(d is not None and d.thr16 > 0 for d in drops) and B0 * h0 * Lq0 * Lk0 <= (1 << 24):
        with torch.cuda.stream(aux):
            for i in range(0, len(sites), 8):
                bits_l = ops.attention_dropmask_multi(drops[i:i + 8], B0, h0, Lq0, Lk0, device)
                ev = torch.cuda.Event()
                ev.record(aux)
                for d, bits in zip(drops[i:i + 8], bits_l):
                    bits.record_stream(main)
                    out.append((d, bits, ev))
        return out
    with torch.cuda.stream(aux):
        for (m, B, Lq, Lk), d in zip(sites, drops):
            bits = ops.attention_dropmask(d, B, m.n_head, Lq, Lk, device)
            if bits is not None:
                bits.record_stream(main)
            ev = torch.cuda.Event()
            ev.record(aux)
            out.append((d, bits, ev))
    return out


# ---- backward tape ---------------------------------------------------------------------------------------------
# The forward of every block pushes one closure that, run in reverse order, turns the gradient of the block's output Act
# into parameter gradients (written straight into `p.grad`, which the trainer points into one flat buffer) and input
# gradients - all through the HIP kernels of backward.hip / attention_bwd.hip / gemm.hip.  Recording needs bf16 precision.
_TAPE = None
_DIRECT_CONV_BWD = True     # the conv layers' direct backward kernels (bf16); the patch-matrix route stays as the f32 parity path
_CONV_F32 = 0               # 1: the conv front end (forward and backward) as an f32 island inside the bf16 step (tools/s2_grad_err.py)
_DECODE_FUSED = True      # one-launch sub-layers in the per-token decode steps (decode_blocks.hip)
_IN_DECODE_STEP = False    # set while a per-token decode step is being queued / captured (rows = hypotheses, one position each)


@contextlib.contextmanager
def _decode_step():
    global _IN_DECODE_STEP
    prev, _IN_DECODE_STEP = _IN_DECODE_STEP, True
    try:
        yield
    finally:
        _IN_DECODE_STEP = prev


_PARAM_EPOCH = 0   # bumped by the trainer after each fused Adam step (raw-pointer updates do not bump tensor versions)


def bump_param_epoch():
    global _PARAM_EPOCH
    _PARAM_EPOCH += 1


class Tape:
    def __init__(self):
        self.fns = []
        self.consumed = False

    def push(self, fn, params=()):
        fn.params = tuple(params)
        self.fns.append(fn)

    def backward(self, after_each=None):
        if self.consumed:     # the closures free their saved activations as they run: like torch without retain_graph, one replay only
            raise RuntimeError("asr_amd: this forward's backward tape has already been replayed (a second backward through the same "
                               "forward is not supported: sum the losses and call backward once)")
        self.consumed = True
        for fn in reversed(self.fns):
            fn()
            if after_each is not None:
                after_each(fn)
        self.fns = []


@contextlib.contextmanager
def record():
    global _TAPE
    old, t = _TAPE, Tape()      # (bf16: the product path; f32: the parity mode - exact-f32 GEMMs, VALU attention, eval mode / dropout 0)
    _TAPE = t
    try:
        yield t
    finally:
        _TAPE = old


def _ln_bwd(*a, **kw):
    """ops.add_layernorm_bwd -> (ds f32, the backward GEMMs' operand): its bf16 image on the product path, ds itself in the fp32 parity mode"""
    ds, ds16 = ops.add_layernorm_bwd(*a, want_bf16=(_PRECISION == "bf16"), **kw)
    return ds, (ds16 if ds16 is not None else ds)


def _add_into(dst, src):
    """dst += src through asr_add2d (f32 CUDA tensors with contiguous rows); anything else falls to torch (CPU-side tests)."""
    if (dst.is_cuda and dst.dtype == torch.float32 and src.dtype == torch.float32 and dst.is_contiguous() and src.dim() >= 1 and
            src.stride(-1) == 1 and dst.shape == src.shape and (src.is_contiguous() or src.dim() == 2)):
        return ops.add_(dst, src)
    return dst.add_(src)


def _acc(act, g):
    act.grad = g if act.grad is None else _add_into(act.grad, g)


def _gcat(params):
    """Gradient view covering adjacent parameters (the trainer lays Q/K/V weights out contiguously)."""
    if len(params) == 1:
        return params[0].grad
    p0 = params[0]
    off = p0._asr_off
    n = 0
    for q in params:
        assert q._asr_off == off + n, "parameters are not adjacent in the flat buffer"
        n += q.numel()
    rows = sum(q.shape[0] for q in params)
    return p0._asr_gflat[off:off + n].view(rows, *p0.shape[1:])


class _Cached(nn.Module):
    """Derived weights (concatenated / re-laid-out / bf16 copies) cached against parameter versions."""

    def _derived(self, key, params, build):
        cache = self.__dict__.setdefault("_wcache", {})
        ver = tuple((p.data_ptr(), p._version) for p in params) + (_PRECISION, _PARAM_EPOCH)
        hit = cache.get(key)
        if hit is None or hit[0] != ver:
            with torch.no_grad():
                hit = (ver, build())
            cache[key] = hit
        return hit[1]

    def _w(self, key, params, dim=0):
        """Compute-dtype copy of (the concatenation of) weight matrices.  When the trainer has re-homed the parameters into
        its flat buffers the bf16 shadow view is returned directly (kept fresh by the fused Adam kernel)."""
        p0 = params[0]
        if _PRECISION == "bf16" and getattr(p0, "_asr_flat16", None) is not None:
            off, n, ok = p0._asr_off, 0, True
            for q in params:
                ok = ok and getattr(q, "_asr_off", -1) == off + n
                n += q.numel()
            if ok:
                rows = sum(q.shape[0] for q in params)
                return p0._asr_flat16[off:off + n].view(rows, *p0.shape[1:])

        def build():
            w = params[0] if len(params) == 1 else torch.cat(list(params), dim)
            w = w.detach().contiguous()
            return ops.cast_bf16(w) if _PRECISION == "bf16" else w.float()
        return self._derived(key, params, build)

    def _b(self, key, params):
        """fp32 (concatenation of) bias vectors; under the trainer adjacent biases are one view of the flat master buffer (no
        per-step torch.cat: the fused Adam kernel updates the master in place)."""
        p0 = params[0]
        flat = getattr(p0, "_asr_flat32", None)
        if flat is not None:
            off, n, ok = p0._asr_off, 0, True
            for q in params:
                ok = ok and getattr(q, "_asr_off", -1) == off + n and getattr(q, "_asr_flat32", None) is flat
                n += q.numel()
            if ok:
                return flat[off:off + n]

        def build():
            b = params[0] if len(params) == 1 else torch.cat(list(params), 0)
            return b.detach().float().contiguous()
        return self._derived(key, params, build)


class Act:
    """An activation as it travels between kernels: fp32 master [M,D] (+ optional bf16 shadow for MFMA)."""
    __slots__ = ("f32", "b16", "B", "L", "grad", "needs_grad", "next_q", "ln_ctx", "ln_done", "lazy_grad")

    def __init__(self, f32, b16, B, L):
        self.f32, self.b16, self.B, self.L, self.grad, self.needs_grad = f32, b16, B, L, None, False
        # a LayerNorm output whose ONLY reader may run that LayerNorm's backward in its own data-gradient launch (asr_ffn_bwd_ln):
        # ln_ctx = what the backward needs (set by the producer), ln_done = (ds, ds16) once a reader has done it
        self.ln_ctx = self.ln_done = None
        self.lazy_grad = None    # a callable that adds a gradient contribution computed elsewhere (the trainer's CTC side branch) into .grad
        self.next_q = None       # decode step: the next cross attention's projected queries, when the launch that made this produced them

    def mma(self):
        return self.b16 if (self.b16 is not None and _PRECISION == "bf16") else self.f32

    def view3(self):
        return self.f32.view(self.B, self.L, -1)


def _act(x):
    B, L, D = x.shape
    return Act(x.contiguous().float().view(B * L, D), None, B, L)


# ------------------------------------------------------------------------------------------------------------
class PositionalEncoding(nn.Module):
    """src/transformer/module.py:7-32 — `pe` is a state_dict buffer (1, max_len, d_model)."""

    def __init__(self, d_model, max_len=5000):
        super().__init__()
        pe = torch.zeros(max_len, d_model, requires_grad=False)
        position = torch.arange(0, max_len).unsqueeze(1).float()
        div_term = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
        pe[:, 0::2] = torch.sin(position * div_term)
        pe[:, 1::2] = torch.cos(position * div_term)
        self.register_buffer("pe", pe.unsqueeze(0))

    def forward(self, input):
        return self.pe[:, :input.size(1)]

    def rows(self, length):
        return self.pe[0, :length].contiguous()


class MultiheadAttention(_Cached):
    """src/transformer/attention.py:6-62 (with ScaledDotProductAttention :65-86 fused in)."""

    def __init__(self, d_model, n_head, d_k=64, d_v=64, dropout=0.1):
        super().__init__()
        if d_k != 64 or d_v != 64:
            raise NotImplementedError("the MFMA attention kernel is specialised for d_k = d_v = 64 "
                                      "(the only head width the reference ever constructs, SURVEY.md)")
        self.n_head, self.d_k, self.d_v = n_head, d_k, d_v
        self.w_qs = nn.Linear(d_model, n_head * d_k)
        self.w_ks = nn.Linear(d_model, n_head * d_k)
        self.w_vs = nn.Linear(d_model, n_head * d_v)
        nn.init.normal_(self.w_qs.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_ks.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_k)))
        nn.init.normal_(self.w_vs.weight, mean=0, std=math.sqrt(2.0 / (d_model + d_v)))
        self.layer_norm = nn.LayerNorm(d_model)
        self.fc = nn.Linear(n_head * d_v, d_model)
        nn.init.xavier_normal_(self.fc.weight)
        self.dropout_rate = dropout

    def _impl(self, xq, xkv, k_len, causal, row_len, kv_pre=None, attn_drop=None, q_pre=None, tail=None):
        """xq: Act [B*Lq, d]; xkv: Act (same object for self-attention).  Returns Act.
        tail(ctx2, wfc, dp_fc, rec) -> proj_ln's tuple or None: the caller's launch that takes this sub-layer's tail with it (EncoderLayer).
        kv_pre = (k, v, dkv) when the caller has already projected the keys / values (_CrossKV: one GEMM for all decoder layers);
        dkv() -> (dk, dv) views the backward writes into, the K/V weight and input gradients are then the caller's business."""
        h, B, Lq, Lk = self.n_head, xq.B, xq.L, xkv.L
        scale = 1.0 / math.sqrt(self.d_k)
        qscale = scale * _LOG2E      # the attention kernels take base-2 logits (asr_hip.h): log2(e) rides on the Q projection
        if xkv is xq:
            W = self._w("qkv", (self.w_qs.weight, self.w_ks.weight, self.w_vs.weight))
            bias = self._b("bqkv", (self.w_qs.bias, self.w_ks.bias, self.w_vs.bias))
            qkv = ops.proj_heads(xq.mma(), W, bias, 3, B, Lq, h, qscale)
            q, k, v = qkv[0], qkv[1], qkv[2]
        else:
            q = q_pre if q_pre is not None else ops.proj_heads(xq.mma(), self._w("q", (self.w_qs.weight,)), self._b("bq", (self.w_qs.bias,)), 1, B, Lq, h, qscale)[0]
            if kv_pre is not None:
                k, v = kv_pre[0], kv_pre[1]
            else:
                kv = ops.proj_heads(xkv.mma(), self._w("kv", (self.w_ks.weight, self.w_vs.weight)),
                                    self._b("bkv", (self.w_ks.bias, self.w_vs.bias)), 2, B, Lk, h, 1.0)
                k, v = kv[0], kv[1]
        rec = _TAPE is not None
        dp_fc = _drop(self, "dropout")   # attention.py:59
        if attn_drop is not None:        # (asr_dropout_t, keep bits, event): hashed ahead of time on a side stream (Encoder._attn_masks)
            dp_attn, dbits, ev = attn_drop
            if ev is not None:
                torch.cuda.current_stream().wait_event(ev)
        else:
            dp_attn = _drop(self, "attention.dropout")   # attention.py:83
            dbits = ops.attention_dropmask(dp_attn, B, h, Lq, Lk, q.device)      # hashed once; forward, dQ and dK/dV kernels read bits
        ctx, lse = ops.attention_fwd(q, k, v, k_len, causal, need_lse=rec, drop=dp_attn, drop_bits=dbits)
        ctx2, wfc = ctx.view(B * Lq, h * 64), self._w("fc", (self.fc.weight,))
        if _PRECISION == "bf16" and ops.gemm_add_layernorm_ok(ctx2, wfc, xq.f32.shape[1], B, Lq):      # projection + LayerNorm in one launch
            o, y32, y16, mean, rstd = ops.gemm_add_layernorm(ctx2, wfc, self._b("bfc", (self.fc.bias,)), xq.f32, self.layer_norm.weight,
                                                                   self.layer_norm.bias, B, Lq, row_len=row_len, eps=self.layer_norm.eps,
                                                                   save_stats=rec, drop_x=dp_fc)
        elif _PRECISION == "bf16" and ops.proj_ln_ok(ctx2, wfc, xq.f32.shape[1], B, Lq) and tail is not None and (fused := tail(ctx2, wfc, dp_fc, rec)) is not None:
            o, y32, y16, mean, rstd = fused      # (this sub-layer's tail ran in the caller's launch: asr_attn_ffn_fwd)
        elif _PRECISION == "bf16" and ops.proj_ln_ok(ctx2, wfc, xq.f32.shape[1], B, Lq):      # the same at encoder size (csrc/ffn.hip, PROJ)
            o, y32, y16, mean, rstd = ops.proj_ln(ctx2, wfc, self._b("bfc", (self.fc.bias,)), xq.f32, self.layer_norm.weight,
                                                  self.layer_norm.bias, B, Lq, row_len=row_len, eps=self.layer_norm.eps, save_stats=rec,
                                                  drop_x=dp_fc, save_s=not _LN_FROM_Y)
        else:
            o = ops.gemm_nt(ctx2, wfc, self._b("bfc", (self.fc.bias,)))
            y32, y16, mean, rstd = ops.add_layernorm(o, xq.f32, self.layer_norm.weight, self.layer_norm.bias, B, Lq, row_len=row_len,
                                                     want_bf16=(_PRECISION == "bf16"), eps=self.layer_norm.eps, save_stats=rec,
                                                     drop_x=dp_fc)
        y = Act(y32, y16, B, Lq)
        if rec:
            self._record_bw(xq, xkv, q, k, v, ctx, lse, o, mean, rstd, y, k_len, causal, row_len, scale, dp_attn, dp_fc,
                            None if kv_pre is None else kv_pre[2], dbits)
        return y

    def _record_bw(self, xq, xkv, q, k, v, ctx, lse, s_sum, mean, rstd, y, k_len, causal, row_len, scale, dp_attn, dp_fc,
                   dkv_pre=None, dbits=None):
        h, B, Lq, Lk = self.n_head, xq.B, xq.L, xkv.L
        hd = h * 64
        ln, fc = self.layer_norm, self.fc
        qkvw = (self.w_qs.weight, self.w_ks.weight, self.w_vs.weight)
        qkvb = (self.w_qs.bias, self.w_ks.bias, self.w_vs.bias)

        from_y = s_sum is None         # (the forward kept no pre-norm sum: x^ comes from the output, ops.add_layernorm_bwd(beta=))
        ln_in, ln_beta = (y.f32, ln.bias) if from_y else (s_sum, None)
        y.ln_ctx = (ln_in, mean, rstd, ln, row_len, fc.bias, dp_fc, ln_beta)

        def bw():
            if y.ln_done is not None:          # the feed-forward sub-layer behind this one ran the LayerNorm's backward in its own launch
                ds, ds16 = y.ln_done
                y.ln_done = None
            else:
                ds, ds16 = _ln_bwd(y.grad, ln_in, mean, rstd, ln.weight, row_len, B, Lq, ln.weight.grad, ln.bias.grad,
                                      dbias=fc.bias.grad, drop_x=dp_fc, beta=ln_beta)
            y.grad = None
            _wg(ds16, ctx.view(B * Lq, hd), out=fc.weight.grad, accumulate=True)
            gdt = _cdtype()
            d_ctx = ops.gemm_nn(ds16, self._w("fc", (fc.weight,)), out_dtype=gdt)
            if xkv is xq:
                dqkv = torch.empty((B * Lq, 3 * hd), device=ds.device, dtype=gdt)
                ops.attention_bwd(q, k, v, ctx, d_ctx, lse, k_len, causal, scale, dqkv[:, :hd], dqkv[:, hd:2 * hd], dqkv[:, 2 * hd:],
                                  drop=dp_attn, drop_bits=dbits)
                _wg(dqkv, xq.mma(), out=_gcat(qkvw), accumulate=True, colsum=_gcat(qkvb))
                wqkv = self._w("qkv", qkvw)
                if _FOLD_LN_QKV and xq.ln_ctx is not None and xq.grad is None and ops.dgrad_rows_ok(dqkv, wqkv):
                    # xq is a LayerNorm output that only this sub-layer read (encoder.py:74-77): its gradient goes through that
                    # LayerNorm's backward in the data-gradient launch itself (asr_dgrad_rows_ln) - no dx in memory, no launch of its own
                    p_s, p_mean, p_rstd, p_ln, p_len, p_bias, p_drop, p_beta = xq.ln_ctx
                    xq.ln_done = ops.gemm_nn_ln(dqkv, wqkv, ds, B, Lq, p_s, p_mean, p_rstd, p_ln.weight, p_len, p_ln.weight.grad,
                                                p_ln.bias.grad, dbias=p_bias.grad, drop_x=p_drop, ln_beta=p_beta)
                else:
                    _acc(xq, ops.gemm_nn(dqkv, wqkv, addend=ds))
            else:
                dq = torch.empty((B * Lq, hd), device=ds.device, dtype=gdt)
                if dkv_pre is not None:
                    dk_out, dv_out = dkv_pre()
                else:
                    dkv = torch.empty((B * Lk, 2 * hd), device=ds.device, dtype=gdt)
                    dk_out, dv_out = dkv[:, :hd], dkv[:, hd:]
                ops.attention_bwd(q, k, v, ctx, d_ctx, lse, k_len, causal, scale, dq, dk_out, dv_out, drop=dp_attn, drop_bits=dbits)
                _wg(dq, xq.mma(), out=self.w_qs.weight.grad, accumulate=True, colsum=self.w_qs.bias.grad)
                _acc(xq, ops.gemm_nn(dq, self._w("q", (self.w_qs.weight,)), addend=ds))
                if dkv_pre is None:
                    _wg(dkv, xkv.mma(), out=_gcat(qkvw[1:]), accumulate=True, colsum=_gcat(qkvb[1:]))
                    xkv.grad = ops.gemm_nn(dkv, self._w("kv", qkvw[1:]), addend=xkv.grad)

        mine = (qkvw[:1] + qkvb[:1]) if dkv_pre is not None else (qkvw + qkvb)
        _TAPE.push(bw, mine + (fc.weight, fc.bias, ln.weight, ln.bias))

    def _impl_ctx(self, xq, xkv, k_len, kv_pre, q=None):
        """decode step only (eval, no tape): the attention output [B * Lq, h * 64] BEFORE the output projection - the caller runs
        projection + residual + LayerNorm inside the next launch (PositionwiseFeedForward._impl_after_attention); q: the projected
        queries when the launch in front has produced them already"""
        h, B, Lq = self.n_head, xq.B, xq.L
        qscale = _LOG2E / math.sqrt(self.d_k)
        if q is None:
            q = ops.proj_heads(xq.mma(), self._w("q", (self.w_qs.weight,)), self._b("bq", (self.w_qs.bias,)), 1, B, Lq, h, qscale)[0]
        ctx, _ = ops.attention_fwd(q, kv_pre[0], kv_pre[1], k_len, False, need_lse=False, drop=None, drop_bits=None)
        return ctx.view(B * Lq, h * 64)

    def _impl_cached_self(self, x, k_cache, v_cache, t, k_len, next_attn=None, next_lq=1):
        """Self-attention of ONE new position t (x: Act [B*1, d]) against the cache (decoding): its key / value are projected and
        written into k_cache / v_cache [B, h, Tmax, 64] at position t, then the query attends to positions < k_len (= t + 1).
        bf16, d_model = 256, position on the device: the whole sub-layer is ONE launch (asr_decode_self_attn)."""
        h, B = self.n_head, x.B
        # (eval semantics only: the one-launch form has no dropout, like _decode_cross_ffn's; a model left in train() takes the
        # separate launches, whose _drop sites stay active in every sub-layer alike)
        if (_DECODE_FUSED and _PRECISION == "bf16" and not self.training and torch.is_tensor(t) and x.L == 1 and
                k_cache.dtype == torch.bfloat16 and self.w_qs.weight.shape[0] == h * 64 and ops.decode_blocks_ok(x, heads=h)):
            # next_attn: the cross-attention module that follows - its query projection of the rows this launch normalises rides along
            # (next_lq rows per utterance are the queries of one cross attention); the result is handed over in `y.next_q`
            nq = None
            if next_attn is not None and next_attn.w_qs.weight.shape == (256, 256) and next_attn.n_head == 4:
                nq = (next_attn._w("q", (next_attn.w_qs.weight,)), next_attn._b("bq", (next_attn.w_qs.bias,)), int(next_lq),
                      _LOG2E / math.sqrt(next_attn.d_k))
            out = ops.decode_self_attn(x.b16 if x.b16 is not None else ops.cast_bf16(x.f32), x.f32, self._w("qkv", (self.w_qs.weight, self.w_ks.weight, self.w_vs.weight)),
                                       self._b("bqkv", (self.w_qs.bias, self.w_ks.bias, self.w_vs.bias)), self._w("fc", (self.fc.weight,)),
                                       self._b("bfc", (self.fc.bias,)), self.layer_norm.weight, self.layer_norm.bias, k_cache, v_cache, t,
                                       self.layer_norm.eps, next_q=nq)
            y = Act(out[0], out[1], B, 1)
            if nq is not None:
                y.next_q = out[2]
            return y
        kv = ops.proj_heads(x.mma(), self._w("kv", (self.w_ks.weight, self.w_vs.weight)), self._b("bkv", (self.w_ks.bias, self.w_vs.bias)),
                            2, B, 1, h, 1.0)
        if torch.is_tensor(t):                      # position in device memory (Decoder.batch_decode's captured step)
            ops.kv_cache_put(kv[0], kv[1], k_cache, v_cache, t)
        else:
            k_cache[:, :, t:t + 1].copy_(kv[0])
            v_cache[:, :, t:t + 1].copy_(kv[1])
        return self._impl(x, Act(None, None, B, k_cache.shape[2]), k_len, False, None, kv_pre=(k_cache, v_cache, None))

    def forward(self, q, k, v, mask=None, k_len=None, causal=False):
        """Reference signature + length-based masking: `k_len` (int [B]) / `causal`.  A bool `mask` [B,Lq,Lk] is
        accepted when it is a key-padding mask (tail padding): it is reduced to k_len.  Returns (output, None)."""
        if mask is not None and k_len is None:
            mb = mask.bool()
            k_len = (~mb[:, -1, :]).sum(-1)
            Lq_, Lk_ = mb.shape[1], mb.shape[2]
            tail = torch.arange(Lk_, device=mb.device)[None, None, :] >= k_len[:, None, None]
            if bool((mb == tail).all()):
                pass                                            # pure key-padding mask (tail padding)
            elif Lq_ == Lk_ and bool((mb == (tail | torch.triu(torch.ones(Lq_, Lk_, dtype=torch.bool, device=mb.device), 1)[None])).all()):
                causal = True                                   # key-padding | subsequent (decoder.py:74-78)
            else:
                raise NotImplementedError("attention mask is neither a tail key-padding mask nor key-padding | causal: pass k_len / causal")
        xq = _act(q)
        xkv = xq if (k is q and v is q) else _act(k)
        if k is not v:
            raise NotImplementedError("key and value must be the same tensor (all reference call sites)")
        kl = None if k_len is None else ops.as_i32(k_len, q.device)
        out = self._impl(xq, xkv, kl, causal, None)
        return out.view3(), None


class PositionwiseFeedForward(_Cached):
    """src/transformer/module.py:35-53."""

    def __init__(self, d_model, d_ff, dropout=0.1):
        super().__init__()
        self.w_1 = nn.Linear(d_model, d_ff)
        self.w_2 = nn.Linear(d_ff, d_model)
        self.layer_norm = nn.LayerNorm(d_model)
        self.dropout_rate = dropout

    def _impl(self, x, row_len, drawn=False):
        """drawn: this call's dropout descriptor when the caller has drawn it already (False: not drawn; None is a drawn 'inactive')"""
        hdt = _cdtype()
        rec = _TAPE is not None
        d_ff = self.w_1.weight.shape[0]
        if (_DECODE_FUSED and not rec and _PRECISION == "bf16" and row_len is None and not self.training and _IN_DECODE_STEP and
                ops.decode_blocks_ok(x, d_ff=d_ff)):
            # the per-token decode step: the whole sub-layer in ONE launch (asr_decode_ffn)
            y32, y16 = ops.decode_ffn(x.b16 if x.b16 is not None else ops.cast_bf16(x.f32), x.f32, self._w("w1", (self.w_1.weight,)), self._b("b1", (self.w_1.bias,)),
                                      self._w("w2", (self.w_2.weight,)), self._b("b2", (self.w_2.bias,)), self.layer_norm.weight,
                                      self.layer_norm.bias, self.layer_norm.eps)
            return Act(y32, y16, x.B, x.L)
        if _PRECISION == "bf16" and ops.ffn_fused_ok(x.b16, x.f32, self._w("w1", (self.w_1.weight,)), self._w("w2", (self.w_2.weight,)), x.B, x.L):
            return self._impl_fused(x, row_len, rec, drawn)
        # training: the ReLU mask travels to the backward as 1 sign bit per hidden unit (written by this GEMM's epilogue) - the
        # hidden gradient's GEMM then reads 8 MB instead of re-reading the 131 MB activation (S1 shape)
        use_bits = rec and _PRECISION == "bf16" and d_ff % 128 == 0 and x.mma().dtype == torch.bfloat16 and x.f32.shape[1] % 64 == 0
        bits = ops.relu_bits_buffer(x.mma().shape[0], d_ff, x.f32.device) if use_bits else None
        if use_bits:
            hid = ops.gemm_nt_ex(x.mma(), self._w("w1", (self.w_1.weight,)), self._b("b1", (self.w_1.bias,)), out_dtype=hdt, relu=True,
                                 relu_bits_out=bits)
        else:
            hid = ops.gemm_nt(x.mma(), self._w("w1", (self.w_1.weight,)), self._b("b1", (self.w_1.bias,)), out_dtype=hdt, relu=True)
        dp = _drop(self, "dropout") if drawn is False else drawn   # module.py:51
        w2m = self._w("w2", (self.w_2.weight,))
        if _PRECISION == "bf16" and ops.gemm_add_layernorm_ok(hid, w2m, x.f32.shape[1], x.B, x.L):       # projection + LayerNorm in one launch
            o, y32, y16, mean, rstd = ops.gemm_add_layernorm(hid, w2m, self._b("b2", (self.w_2.bias,)), x.f32, self.layer_norm.weight,
                                                                   self.layer_norm.bias, x.B, x.L, row_len=row_len, eps=self.layer_norm.eps,
                                                                   save_stats=rec, drop_x=dp)
        else:
            o = ops.gemm_nt(hid, w2m, self._b("b2", (self.w_2.bias,)))
            y32, y16, mean, rstd = ops.add_layernorm(o, x.f32, self.layer_norm.weight, self.layer_norm.bias, x.B, x.L, row_len=row_len,
                                                     want_bf16=(_PRECISION == "bf16"), eps=self.layer_norm.eps, save_stats=rec, drop_x=dp)
        y = Act(y32, y16, x.B, x.L)
        if rec:
            ln, w1, w2 = self.layer_norm, self.w_1, self.w_2

            def bw():
                ds, ds16 = _ln_bwd(y.grad, o, mean, rstd, ln.weight, row_len, x.B, x.L, ln.weight.grad, ln.bias.grad,
                                      dbias=w2.bias.grad, drop_x=dp)
                y.grad = None
                _wg(ds16, hid, out=w2.weight.grad, accumulate=True)
                d_hid = (ops.gemm_nn(ds16, self._w("w2", (w2.weight,)), out_dtype=_cdtype(), relu_bits=bits) if bits is not None
                         else ops.gemm_nn(ds16, self._w("w2", (w2.weight,)), out_dtype=_cdtype(), relu_mask=hid))
                _wg(d_hid, x.mma(), out=w1.weight.grad, accumulate=True, colsum=w1.bias.grad)
                _acc(x, ops.gemm_nn(d_hid, self._w("w1", (w1.weight,)), addend=ds))

            _TAPE.push(bw, (w1.weight, w1.bias, w2.weight, w2.bias, ln.weight, ln.bias))
        return y

    def _impl_fused(self, x, row_len, rec, drawn=False):
        """Encoder-sized rows at d_model = 256: the sub-layer is ONE forward launch (asr_ffn_fwd: both products, bias, ReLU, dropout,
        residual, LayerNorm, row mask; the hidden activation is written once for the weight gradient, never read back in the forward)
        and ONE data-gradient launch (asr_ffn_bwd: dH and dX); the two weight gradients stay GEMMs over the stored H / dH."""
        ln, w1, w2 = self.layer_norm, self.w_1, self.w_2
        w1m, w2m = self._w("w1", (w1.weight,)), self._w("w2", (w2.weight,))
        dp = _drop(self, "dropout") if drawn is False else drawn   # module.py:51
        outs = ops.ffn_fwd(x.b16, x.f32, w1m, self._b("b1", (w1.bias,)), w2m, self._b("b2", (w2.bias,)), ln.weight, ln.bias, x.B, x.L,
                           row_len=row_len, eps=ln.eps, train=rec, drop_x=dp, save_s=not _LN_FROM_Y)
        return self._fused_finish(x, row_len, rec, dp, outs)

    def _fused_finish(self, x, row_len, rec, dp, outs):
        """the sub-layer's output Act and backward closure from asr_ffn_fwd's tensors (ops.ffn_fwd, or ops.attn_ffn_fwd's second tuple)"""
        ln, w1, w2 = self.layer_norm, self.w_1, self.w_2
        hid, bits, o, y32, y16, mean, rstd = outs
        y = Act(y32, y16, x.B, x.L)
        if rec:
            ln_in, ln_beta = (y32, ln.bias) if o is None else (o, None)
            y.ln_ctx = (ln_in, mean, rstd, ln, row_len, w2.bias, dp, ln_beta)

            def bw():
                if y.ln_done is not None:      # the next layer's attention sub-layer ran this LayerNorm's backward in its data-gradient launch
                    ds, ds16 = y.ln_done
                    y.ln_done = None
                else:
                    ds, ds16 = _ln_bwd(y.grad, ln_in, mean, rstd, ln.weight, row_len, x.B, x.L, ln.weight.grad, ln.bias.grad,
                                          dbias=w2.bias.grad, drop_x=dp, beta=ln_beta)
                y.grad = None
                fold = _FOLD_LN and x.ln_ctx is not None and x.grad is None and _PRECISION == "bf16"
                if fold:
                    # x is a LayerNorm output that only this sub-layer read (encoder.py:74-76): dx goes through that LayerNorm's
                    # backward in ffn_bwd's epilogue instead of through memory and a launch of its own
                    p_s, p_mean, p_rstd, p_ln, p_len, p_bias, p_drop, p_beta = x.ln_ctx
                    d_hid, p_ds, p_ds16 = ops.ffn_bwd_ln(ds16, ds, self._w("w1", (w1.weight,)), self._w("w2", (w2.weight,)), bits, x.B, x.L,
                                                         p_s, p_mean, p_rstd, p_ln.weight, p_len, p_ln.weight.grad, p_ln.bias.grad,
                                                         dbias=p_bias.grad, drop_x=p_drop, ln_beta=p_beta)
                    x.ln_done = (p_ds, p_ds16)
                else:
                    d_hid, dx = ops.ffn_bwd(ds16, ds, self._w("w1", (w1.weight,)), self._w("w2", (w2.weight,)), bits)
                # (both weight gradients behind the data gradient: the side stream then needs ONE event of the main chain per sub-layer,
                # recorded after ffn_bwd - an event recorded between LayerNorm backward and ffn_bwd held the latter back ~5 us)
                _wg(ds16, hid, out=w2.weight.grad, accumulate=True)
                _wg(d_hid, x.mma(), out=w1.weight.grad, accumulate=True, colsum=w1.bias.grad)
                if not fold:
                    _acc(x, dx)

            _TAPE.push(bw, (w1.weight, w1.bias, w2.weight, w2.bias, ln.weight, ln.bias))
        return y

    def _impl_after_attention(self, ctx16, attn, x_in):
        """decode step only: LayerNorm_attn(ctx . Wfc^T + bfc + x_in) and this feed-forward sub-layer in ONE launch (asr_decode_ffn
        with its prologue); ctx16 bf16 [M, 256] is `attn`'s attention output, x_in the Act that entered the attention sub-layer"""
        pre = (attn._w("fc", (attn.fc.weight,)), attn._b("bfc", (attn.fc.bias,)), attn.layer_norm.weight, attn.layer_norm.bias, attn.layer_norm.eps)
        y32, y16 = ops.decode_ffn(ctx16, x_in.f32, self._w("w1", (self.w_1.weight,)), self._b("b1", (self.w_1.bias,)),
                                  self._w("w2", (self.w_2.weight,)), self._b("b2", (self.w_2.bias,)), self.layer_norm.weight,
                                  self.layer_norm.bias, self.layer_norm.eps, pre=pre)
        return Act(y32, y16, x_in.B, x_in.L)

    def forward(self, x):
        return self._impl(_act(x), None).view3()


class EncoderLayer(nn.Module):
    """src/transformer/encoder.py:61-79."""

    def __init__(self, d_model, d_inner, n_head, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiheadAttention(d_model, n_head, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def _impl(self, x, lens, causal=False, attn_drop=None):
        att, ffn = self.slf_attn, self.pos_ffn
        box = {}

        def tail(ctx2, wfc, dp_fc, rec):
            """the attention sub-layer's output projection + residual + LayerNorm and the whole feed-forward sub-layer in ONE launch
            (asr_attn_ffn_fwd: both own the same 128-token row blocks) -> proj_ln's tuple; the feed-forward tensors wait in `box`"""
            w1m, w2m = ffn._w("w1", (ffn.w_1.weight,)), ffn._w("w2", (ffn.w_2.weight,))
            ffn_drops = bool(ffn.training and float(getattr(ffn, "dropout_rate", 0.0) or 0.0) > 0.0)
            if not ops.attn_ffn_ok(ctx2, wfc, x.f32, w1m, w2m, x.B, x.L) or ffn_drops != (dp_fc is not None):
                return None
            dp = _drop(ffn, "dropout")   # module.py:51
            if (dp is None) != (dp_fc is None):
                box["dp"] = dp          # (drawn already: the separate launch below takes it)
                return None
            ln0, ln = att.layer_norm, ffn.layer_norm
            pre, main = ops.attn_ffn_fwd(ctx2, wfc, att._b("bfc", (att.fc.bias,)), x.f32, ln0.weight, ln0.bias, w1m, ffn._b("b1", (ffn.w_1.bias,)),
                                         w2m, ffn._b("b2", (ffn.w_2.bias,)), ln.weight, ln.bias, x.B, x.L, row_len=lens, eps0=ln0.eps,
                                         eps=ln.eps, train=rec, drop0=dp_fc, drop_x=dp, save_s=not _LN_FROM_Y)
            box["ffn"] = (dp, main)
            return pre

        y = att._impl(x, x, lens, causal, lens, attn_drop=attn_drop, tail=tail if _PRECISION == "bf16" else None)   # `*= non_pad_mask` fused into the LN kernel
        if "ffn" in box:
            return ffn._fused_finish(y, lens, _TAPE is not None, *box["ffn"])
        return ffn._impl(y, lens, drawn=box.get("dp", False))

    def forward(self, enc_input, non_pad_mask=None, slf_attn_mask=None, lengths=None):
        if lengths is None and non_pad_mask is not None:
            lengths = non_pad_mask.reshape(non_pad_mask.shape[0], -1).sum(-1)
        lens = None if lengths is None else ops.as_i32(lengths, enc_input.device)
        return self._impl(_act(enc_input), lens).view3()


class Encoder(_Cached):
    """src/transformer/encoder.py:8-58."""

    def __init__(self, d_input, n_layers, n_head, d_model, d_inner, dropout=0.1):
        super().__init__()
        self.d_input, self.n_layers, self.n_head = d_input, n_layers, n_head
        self.d_model, self.d_output, self.d_inner, self.dropout_rate = d_model, d_model, d_inner, dropout
        self.linear_in = nn.Linear(d_input, d_model)
        self.layer_norm_in = nn.LayerNorm(d_model)
        self.positional_encoding = PositionalEncoding(d_model)
        self.layer_stack = nn.ModuleList([EncoderLayer(d_model, d_inner, n_head, dropout=dropout) for _ in range(n_layers)])

    def _impl(self, x, lens):
        B, L = x.B, x.L
        rec = _TAPE is not None
        x_in = x
        o = ops.gemm_nt(x.mma(), self._w("lin", (self.linear_in.weight,)), self._b("blin", (self.linear_in.bias,)))
        dp = _drop(self, "dropout")   # encoder.py:48
        y32, y16, mean, rstd = ops.add_layernorm(o, None, self.layer_norm_in.weight, self.layer_norm_in.bias, B, L,
                                                 pe=self.positional_encoding.rows(L), want_bf16=(_PRECISION == "bf16"),
                                                 eps=self.layer_norm_in.eps, save_stats=rec, drop_y=dp)
        x = Act(y32, y16, B, L)
        if rec:
            y0, ln, lin = x, self.layer_norm_in, self.linear_in
            need_dx = x_in.needs_grad

            def bw():
                # the tape's last closure: the weight gradients still waiting for a grouped launch go to the side stream NOW, beside this
                # closure's three launches - flushed at the end of the backward they ran alone, on half the chip, while Adam waited
                # (tools/step_tail.py: main chain's last kernel -> Adam 136 -> 88 us; S1 - 0.02 ms, S2 - 0.02 ms, five pairs each)
                flush_wgrads()
                ds, ds16 = _ln_bwd(y0.grad, o, mean, rstd, ln.weight, None, B, L, ln.weight.grad, ln.bias.grad,
                                      dbias=lin.bias.grad, drop_y=dp)
                y0.grad = None
                ops.gemm_tn(ds16, x_in.mma(), out=lin.weight.grad, accumulate=True)
                if need_dx:
                    _acc(x_in, ops.gemm_nn(ds16, self._w("lin", (lin.weight,))))

            _TAPE.push(bw, (lin.weight, lin.bias, ln.weight, ln.bias))
        masks = self._attn_masks(B, L, y32.device)
        for i, layer in enumerate(self.layer_stack):
            x = layer._impl(x, lens, attn_drop=masks[i])
        return x

    def _attn_masks(self, B, L, device):
        return _prefetch_attn_masks([(layer.slf_attn, B, L, L) for layer in self.layer_stack], self.training, device)

    def forward(self, padded_input, input_lengths):
        lens = ops.as_i32(input_lengths, padded_input.device)
        return self._impl(_act(padded_input), lens).view3()


class Conv2dSubsample(_Cached):
    """src/transformer/conv_encoder.py:81-126 (pad='same')."""

    def __init__(self, d_input, d_model, n_layers=2, pad="same"):
        super().__init__()
        assert n_layers >= 1
        if pad != "same":
            raise NotImplementedError("only pad='same' (the reference default, never overridden)")
        self.n_layers, self.d_input, self.pad = n_layers, d_input, pad
        from collections import OrderedDict
        layers = [("subsample/conv0", nn.Conv2d(1, 32, 3, (2, 1))), ("subsample/relu0", nn.ReLU())]
        for i in range(n_layers - 1):
            layers += [("subsample/conv{}".format(i + 1), nn.Conv2d(32, 32, 3, (2, 1))),
                       ("subsample/relu{}".format(i + 1), nn.ReLU())]
        self.conv = nn.Sequential(OrderedDict(layers))  # parameter container only; compute is in conv.hip
        self.d_conv_out = int(math.ceil(d_input / 2))
        self.affine = nn.Linear(32 * self.d_conv_out, d_model)

    def _conv_params(self, i):
        m = getattr(self.conv, "subsample/conv{}".format(i))
        return m.weight, m.bias

    def _impl(self, feats, feat_lengths):
        if _CONV_F32 and _PRECISION == "bf16":
            with precision("f32"):
                return self._impl(feats, feat_lengths)
        B, T, D = feats.shape
        n = self.n_layers
        F = self.d_conv_out
        tl = T
        for _ in range(n):
            tl = int(math.ceil(tl / 2.0))
        tneed, fneed = [0] * n, [0] * n
        tneed[n - 1], fneed[n - 1] = tl, F
        for i in range(n - 2, -1, -1):
            tneed[i], fneed[i] = 2 * tneed[i + 1] + 1, fneed[i + 1] + 2
        feats = feats.float().contiguous()
        w0, b0 = self._conv_params(0)
        ys = [ops.conv_sub0(feats, w0.detach().float().contiguous(), b0.detach().float().contiguous(), _cdtype(), tneed[0], fneed[0])]
        for i in range(1, n):
            w, b = self._conv_params(i)
            ys.append(ops.conv_sub1(ys[-1], w.detach().float().contiguous(), b.detach().float().contiguous(), tneed[i], fneed[i],
                                    last=False))
        # the last layer stays channel-LAST [B,tl,F,32]; the reference's [B,T,C*D] flattening (conv_encoder.py:108, column c*F+f) is
        # absorbed by permuting the affine weight's columns to f*32+c once (a derived weight)
        aff = self.affine

        def perm_w():
            w = aff.weight.detach().view(-1, 32, F).permute(0, 2, 1).reshape(-1, F * 32).contiguous()
            return ops.cast_bf16(w) if _PRECISION == "bf16" else w.float()

        wp = self._derived("aff_perm", (aff.weight,), perm_w)
        y_last = ys[-1].view(B * tl, F * 32)
        out = ops.gemm_nt(y_last, wp, self._b("baff", (aff.bias,)))
        lens = feat_lengths.to(feats.device)
        for _ in range(n):
            lens = torch.div(lens + 1, 2, rounding_mode="floor")  # == ceil(len / 2) for non-negative ints
        act = Act(out, None, B, tl)
        act.needs_grad = True
        if _TAPE is not None:
            self._record_bw(feats, ys, tneed, fneed, act, wp, y_last)
        return act, lens.to(torch.int32)

    def _record_bw(self, feats, ys, tneed, fneed, act, wp, y_last):
        n, F, aff = self.n_layers, self.d_conv_out, self.affine
        B, T, D = feats.shape
        convs = [getattr(self.conv, "subsample/conv{}".format(i)) for i in range(n)]

        island = _PRECISION == "f32" and _CONV_F32

        def bw():
            if island and _PRECISION == "bf16":
                with precision("f32"):
                    return bw()
            d_out = act.grad
            act.grad = None
            d = d_out.shape[1]
            cdt = _cdtype()
            # [d, F*32] in the permuted column order (bf16 x bf16 operands take the LDS-DMA weight-gradient kernel; with the f32 gradient
            # as it arrives the generic kernel converts on load and is 3x slower at [8000, 256] x [8000, 1280])
            dwp = ops.gemm_tn(ops.cast_bf16(d_out) if (cdt == torch.bfloat16 and y_last.dtype == torch.bfloat16) else d_out, y_last)
            ops.add_transposed_(aff.weight.grad, dwp, d, 32, F)                      # affine weight columns are c*F + f (conv_encoder.py:108)
            ops.colsum(d_out, out=aff.bias.grad, accumulate=True)
            dy = ops.gemm_nn(d_out, wp, out_dtype=cdt, relu_mask=y_last)  # [M, F*32] = channel-last d(y_last), ReLU applied
            for i in range(n - 1, 0, -1):
                cv = convs[i]
                tout, fout = tneed[i], fneed[i]
                dy2 = dy.view(-1, 32)
                xin = ys[i - 1]
                if (_DIRECT_CONV_BWD and cdt == torch.bfloat16 and fout <= 48 and xin.shape[2] <= 50 and xin.shape[2] >= fout + 2 and
                        xin.shape[1] >= 2 * tout + 1):
                    # direct kernels: dy and x read once each (the patch-matrix route below moves 184 MB four times at S2)
                    dwm = ops.conv_sub1_bwd_w(dy2, xin, tout, fout, db=cv.bias.grad)
                    ops.add_transposed_(cv.weight.grad, dwm, 32, 32, 9)
                    dy = ops.conv_sub1_bwd_x(dy2, cv.weight.detach().float().contiguous(), xin, tout, fout)
                    continue
                col = ops.conv_im2col(ys[i - 1], 32, tout, fout, 288, cdt)
                dwm = ops.gemm_tn(dy2, col)                                         # [co, tap*32 + ci]
                ops.add_transposed_(cv.weight.grad, dwm, 32, 32, 9)                 # nn.Conv2d weight is [co, ci, 3, 3]
                ops.colsum(dy2, out=cv.bias.grad, accumulate=True)
                wm = self._derived("wm%d" % i, (cv.weight,),
                                   lambda cv=cv: (ops.cast_bf16 if _PRECISION == "bf16" else (lambda t: t.float()))(
                                       cv.weight.detach().permute(0, 2, 3, 1).reshape(32, 288).contiguous()))
                dcol = ops.gemm_nn(dy2, wm, out_dtype=cdt)
                dy = ops.conv_col2im_relu(dcol, ys[i - 1], tout, fout)              # [B,Tin,Fin,32], masked by relu'(y_{i-1})
            dy2 = dy.view(-1, 32)
            if (_DIRECT_CONV_BWD and cdt == torch.bfloat16 and dy2.dtype == torch.bfloat16 and fneed[0] <= 48 and feats.dtype == torch.float32 and
                    feats.is_contiguous()):
                ops.conv_sub0_bwd_w(dy2, feats, convs[0].weight.grad, convs[0].bias.grad, tneed[0], fneed[0])
                return
            col0 = ops.conv_im2col(feats.view(B, T, D, 1), 1, tneed[0], fneed[0], 12, torch.float32)
            dw0 = ops.gemm_tn(dy2, col0)                                            # [32, 12]; taps in columns 0..8
            ops.add_transposed_(convs[0].weight.grad, dw0, 32, 9, 1, lds=dw0.stride(0))
            ops.colsum(dy2, out=convs[0].bias.grad, accumulate=True)

        params = [aff.weight, aff.bias]
        for cv in convs:
            params += [cv.weight, cv.bias]
        _TAPE.push(bw, params)

    def forward(self, feats, feat_lengths):
        a, lens = self._impl(feats, feat_lengths)
        return a.view3(), lens


class Conv1d(_Cached):
    """src/transformer/conv_encoder.py:9-49 (pad='same').  k=w valid convs expressed as GEMMs over overlapping
    row windows (lda = C, K = w*C) of a right-zero-padded [B, L + n*w, C] buffer."""

    def __init__(self, d_input, d_hidden, n_layers, w_context, pad="same", name=""):
        super().__init__()
        assert n_layers >= 1
        self.n_layers, self.d_input, self.d_hidden, self.w_context, self.pad, self.name = (
            n_layers, d_input, d_hidden, w_context, pad, name)
        from collections import OrderedDict
        layers = [("{}/conv1d_0".format(name), nn.Conv1d(d_input, d_hidden, w_context, 1)), ("{}/relu_0".format(name), nn.ReLU())]
        for i in range(n_layers - 1):
            layers += [("{}/conv1d_{}".format(name, i + 1), nn.Conv1d(d_hidden, d_hidden, w_context, 1)),
                       ("{}/relu_{}".format(name, i + 1), nn.ReLU())]
        self.conv = nn.Sequential(OrderedDict(layers))

    def _wg(self, i):
        m = getattr(self.conv, "{}/conv1d_{}".format(self.name, i))
        return self._derived("w%d" % i, (m.weight,), lambda m=m: (
            ops.cast_bf16(m.weight.detach().permute(0, 2, 1).reshape(m.weight.shape[0], -1).contiguous())
            if _PRECISION == "bf16" else m.weight.detach().permute(0, 2, 1).reshape(m.weight.shape[0], -1).contiguous().float()))

    def _impl(self, x, save=False):
        """x: Act [B*L, C] -> f32 tensor [B, L, d_hidden] (and the per-layer buffers when save=True, for the backward)"""
        B, L, w, n = x.B, x.L, self.w_context, self.n_layers
        Lp = L + n * w
        C = x.f32.shape[1]
        cd = _cdtype()
        buf = torch.zeros((B * Lp + w, C), device=x.f32.device, dtype=cd)
        src = x.mma() if x.mma().dtype == cd else (ops.cast_bf16(x.f32) if cd == torch.bfloat16 else x.f32)
        buf[:B * Lp].view(B, Lp, C)[:, :L].copy_(src.view(B, L, C))
        bufs, outs = [], []
        for i in range(n):
            m = getattr(self.conv, "{}/conv1d_{}".format(self.name, i))
            last = i == n - 1
            out = torch.zeros((B * Lp + w, self.d_hidden), device=buf.device, dtype=torch.float32 if last else cd)
            ops.gemm_nt_raw(buf, B * Lp, w * buf.shape[1], buf.shape[1], self._wg(i), m.bias.detach().float().contiguous(), relu=True,
                            out=out, ldc=self.d_hidden)
            bufs.append(buf)
            outs.append(out)
            buf = out
        y = buf[:B * Lp].view(B, Lp, self.d_hidden)[:, :L].contiguous()
        return (y, (bufs, outs, B, L, Lp)) if save else y

    def _backward(self, saved, d_y):
        """d_y f32 [B, L, d_hidden] -> d_x f32 [B*L, C]; accumulates the conv weights' / biases' gradients."""
        bufs, outs, B, L, Lp = saved
        w, n = self.w_context, self.n_layers
        rows = B * Lp
        d = torch.zeros((rows + w, self.d_hidden), device=d_y.device, dtype=torch.float32)
        d[:rows].view(B, Lp, self.d_hidden)[:, :L].copy_(d_y)
        for i in range(n - 1, -1, -1):
            m = getattr(self.conv, "{}/conv1d_{}".format(self.name, i))
            cin = bufs[i].shape[1]
            d_pre = ops.relu_mask_mul(d, outs[i])                                     # ReLU backward from the saved output [rows + w, d_hidden]
            win = torch.as_strided(bufs[i], (rows, w * cin), (cin, 1))              # the overlapping row windows the forward GEMM read
            dwg = ops.gemm_tn(d_pre[:rows], win)                                      # [O, w*cin] in (tap, channel) column order
            ops.add_transposed_(m.weight.grad, dwg, dwg.shape[0], cin, w)             # nn.Conv1d weight is [O, cin, w]
            ops.colsum(d_pre[:rows], out=m.bias.grad, accumulate=True)
            d_win = ops.gemm_nn(d_pre[:rows], self._wg(i))                            # [rows, w*cin] f32
            d = ops.conv1d_overlap_add(d_win, rows, w, cin)                           # [rows + w, cin]
        return d[:rows].view(B, Lp, -1)[:, :L].reshape(B * L, -1).contiguous()

    def forward(self, feats, feat_lengths):
        return self._impl(_act(feats)), feat_lengths


class Attention_Assigner(nn.Module):
    """src/transformer/attentionAssigner.py:8-40."""

    def __init__(self, d_input, d_hidden, w_context, n_layers, dropout=0.1):
        super().__init__()
        self.d_input, self.d_hidden, self.n_layers, self.w_context = d_input, d_hidden, n_layers, w_context
        self.conv = Conv1d(d_input, d_hidden, n_layers, w_context, pad="same", name="assigner")
        self.linear = nn.Linear(d_hidden, 1)
        self.dropout_rate = dropout

    def _impl(self, x, lens, slot=None):
        """-> alpha f32 [B,L].  When the tape is recording, `slot` ({"g": d_alpha}) is read by the pushed backward closure."""
        rec = _TAPE is not None and slot is not None
        w = self.linear.weight.detach().float().contiguous().view(-1)
        dp = _drop(self, "dropout")   # attentionAssigner.py:35
        if not rec:
            hcv = self.conv._impl(x)
            if dp is not None:
                ops.dropout_apply(hcv, dp, x.B, x.L, hcv.shape[-1])
            return ops.assigner_tail(hcv, w, self.linear.bias.detach().float().contiguous(), lens, x.B, x.L)
        hcv, saved = self.conv._impl(x, save=True)
        if dp is not None:   # hcv is the cropped copy; `saved` keeps the un-dropped activations as the conv stack's ReLU masks
            ops.dropout_apply(hcv, dp, x.B, x.L, hcv.shape[-1])
        alpha = ops.assigner_tail(hcv, w, self.linear.bias.detach().float().contiguous(), lens, x.B, x.L)
        lin, conv = self.linear, self.conv
        params = [lin.weight, lin.bias]
        for i in range(conv.n_layers):
            cm = getattr(conv.conv, "{}/conv1d_{}".format(conv.name, i))
            params += [cm.weight, cm.bias]

        def bw():
            # sigmoid' * Linear(d_h -> 1) backward in one kernel (alpha is 0 on masked frames, so is dz)
            d_hcv = ops.assigner_tail_bwd(slot["g"], alpha, hcv, w, x.B, x.L, lin.weight.grad, lin.bias.grad)
            if dp is not None:
                d_hcv = ops.dropout_apply(d_hcv, dp, x.B, x.L, d_hcv.shape[-1])
            _acc(x, conv._backward(saved, d_hcv.view(x.B, x.L, -1)))
            slot["g"] = None

        _TAPE.push(bw, params)
        return alpha

    def forward(self, padded_input, input_lengths):
        return self._impl(_act(padded_input), ops.as_i32(input_lengths, padded_input.device))


# ------------------------------------------------------------------------------------------------------------
class DecoderLayer(nn.Module):
    """src/transformer/decoder.py:618-639."""

    def __init__(self, d_model, d_inner, n_head, dropout=0.1):
        super().__init__()
        self.slf_attn = MultiheadAttention(d_model, n_head, dropout=dropout)
        self.enc_attn = MultiheadAttention(d_model, n_head, dropout=dropout)
        self.pos_ffn = PositionwiseFeedForward(d_model, d_inner, dropout=dropout)

    def _decode_cross_ffn(self, x, enc, enc_len, kv_pre, n_rows, q=None):
        """the cross-attention and feed-forward sub-layers of one decode step (x: Act [B, Lq] rows, Lq = 1 or the beam) -> Act
        [n_rows, 1]: query projection, attention, then output projection + LayerNorm + feed-forward in one launch when the step's
        rows fit the fused kernel (bf16, d_model 256, <= 64 rows), else the two sub-layers as usual"""
        ffn, att = self.pos_ffn, self.enc_attn
        if (_DECODE_FUSED and _PRECISION == "bf16" and _IN_DECODE_STEP and not ffn.training and att.fc.weight.shape[1] == 256 and
                ops.decode_blocks_ok(x, d_ff=ffn.w_1.weight.shape[0])):
            ctx = att._impl_ctx(x, enc, enc_len, kv_pre, q=q)
            y = ffn._impl_after_attention(ctx, att, x)
            return Act(y.f32, y.b16, n_rows, 1)
        xq = att._impl(x, enc, enc_len, False, None, kv_pre=kv_pre, q_pre=q)
        return ffn._impl(Act(xq.f32, xq.b16, n_rows, 1), None)

    def _impl(self, x, enc, dec_len, enc_len, kv_pre=None, attn_drop=(None, None)):
        x = self.slf_attn._impl(x, x, dec_len, True, dec_len, attn_drop=attn_drop[0])
        x = self.enc_attn._impl(x, enc, enc_len, False, dec_len, kv_pre=kv_pre, attn_drop=attn_drop[1])
        return self.pos_ffn._impl(x, dec_len)


def _vocab_proj(mod, key, weight, x, ctc=None):
    """logits = x . W^T (no bias).  On the tape the gradient arrives through `mod._grad_slots[key]["g"]`, filled by the
    trainer from the fused loss backward: a [.., V] view of a zero-padded buffer whose rows are 16-byte aligned.
    ctc = (targets, input lengths): the training step's CTC branch -> (logits, None or (loss, nll, state)): when the shape takes it
    (ops.vocab_proj_ctc: encoder-sized rows, d_model 256, bf16 mode) the projection's own launch leaves fp16 logits, the rows'
    log-sum-exp and the CTC table rows, and the alpha / beta recursion has already run on them."""
    w16 = mod._w(key, (weight,))
    done = None
    # rows padded to a multiple of 8 floats: every row 16-byte aligned -> the GEMM's full-cache-line vector epilogue applies to any
    # vocabulary size (V = 4234: 295 -> ~150 us for the [32000, V] CTC projection); the loss kernels take the row stride
    xa = x.mma()
    M, V = xa.shape[0], weight.shape[0]
    Vp = (V + 7) // 8 * 8
    if (ctc is not None and _PRECISION == "bf16" and ops.FUSED_VOCAB_CTC and xa.shape[0] >= 4096 and
            ops.vocab_proj_ctc_ok(xa, w16, x.B, x.L, ctc[0].shape[1])):
        logits, c_loss, c_nll, c_st = ops.vocab_proj_ctc(xa, w16, ctc[0], ctc[1], x.B, x.L)
        done = (c_loss, c_nll, c_st)
    else:
        buf = torch.empty((M, Vp), device=xa.device, dtype=torch.float32)
        ops.gemm_nt_raw(xa, M, xa.shape[1], xa.shape[1], w16, None, out=buf, ldc=Vp)
        logits = buf[:, :V]
    if _TAPE is not None:
        slot = {"g": None, "shape": tuple(logits.shape)}
        mod.__dict__.setdefault("_grad_slots", {})[key] = slot

        def bw():
            g = slot["g"]
            V = g.shape[-1]
            rows = g.numel() // V
            # f32, or the fused CTC backward's bf16 image (rows padded with zeros to a multiple of 128: both GEMMs below then run
            # their LDS-DMA kernels on a vocabulary size that is a multiple of nothing)
            ok = (g.dtype in (torch.float32, torch.bfloat16) and g.stride(-1) == 1 and g.stride(-2) % 8 == 0 and g.data_ptr() % 16 == 0 and
                  (g.dim() == 2 or g.stride(0) == g.shape[1] * g.stride(1)))
            if not ok:   # a gradient that did not come from the fused loss kernels: re-home it into a zero-padded, aligned buffer
                buf = torch.zeros((rows, (V + 7) // 8 * 8), device=g.device, dtype=torch.float32)
                buf[:, :V].copy_(g.reshape(rows, V))
                g = buf[:, :V]
            Vp = g.stride(-2)
            g2 = torch.as_strided(g, (rows, V), (Vp, 1), g.storage_offset())
            _wg(g2, x.mma(), out=weight.grad, accumulate=True)      # (under the trainer's backward: with the decoder's other weight gradients, off the main chain)
            x.grad = ops.gemm_nn(g2, w16, addend=x.grad)
            slot["g"] = None

        _TAPE.push(bw, (weight,))
    return (logits, done) if ctc is not None else logits


def _compact_targets(targets):
    """rows with zeros stripped (decoder.py:46) -> (compacted [B,U], count [B])"""
    nz = targets != 0
    n = nz.sum(1)
    order = torch.argsort((~nz).to(torch.int8), dim=1, stable=True)
    comp = torch.gather(targets, 1, order)
    comp = comp * (torch.arange(targets.shape[1], device=targets.device)[None, :] < n[:, None])
    return comp, n


class _DecodeGraph(dict):
    """The buffers, closures and hipGraphs of one Decoder.batch_decode shape.  Dropped in a fixed order, after the device is idle:
    first the replay handles, then every tensor the graphs address (some live in the graphs' private pool), then the graphs."""
    keep = None

    def __del__(self):
        try:
            if self.get("graphs") is not None and not torch.cuda.is_current_stream_capturing():   # (never synchronise inside someone's capture)
                torch.cuda.synchronize()
            self.pop("prologue", None)
            self.pop("step", None)
            graphs = self.pop("graphs", None)
            self.keep = None
            self.clear()
            del graphs
        except Exception:
            pass


class Decoder(_Cached):
    """src/transformer/decoder.py:13-96 (forward path; decode loops are out of scope)."""

    def __init__(self, sos_id, eos_id, n_tgt_vocab, n_layers, n_head, d_model, d_inner, dropout=0.1):
        super().__init__()
        self.sos_id, self.eos_id, self.n_tgt_vocab = sos_id, eos_id, n_tgt_vocab
        self.d_word_vec, self.n_layers, self.n_head = d_model, n_layers, n_head
        self.d_model, self.d_inner, self.d_output, self.dropout_rate = d_model, d_inner, n_tgt_vocab, dropout
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_model)
        self.positional_encoding = PositionalEncoding(d_model)
        self.layer_stack = nn.ModuleList([DecoderLayer(d_model, d_inner, n_head, dropout=dropout) for _ in range(n_layers)])
        self.tgt_word_prj = nn.Linear(d_model, n_tgt_vocab, bias=False)
        nn.init.xavier_normal_(self.tgt_word_prj.weight)

    def preprocess(self, targets, umax=None):
        """decoder.py:42-58 — strip pad(0), prepend <sos> / append <eos>, re-pad with 0.  `umax` = the longest target of the batch
        when the caller knows it (the data loader does): without it one host sync reads it back from the device."""
        return self._preprocess(targets, umax)[:2]

    def _preprocess(self, targets, umax=None):
        """-> (ys_in, ys_out, number of positive entries per row of ys_in as int32)"""
        hint = self.__dict__.get("_pre_hint")      # the trainer pre-computes this before queueing the step (no mid-step host sync)
        if hint is not None and hint[0] is targets:
            if len(hint) > 2 and hint[2] is not None:          # computed on a side stream: order the consumer after it
                torch.cuda.current_stream().wait_event(hint[2])
            return hint[1]
        if targets.is_cuda and targets.dtype == torch.int64:
            if umax is None:
                umax = int((targets != 0).sum(1).max().item())
            # a caller-supplied `umax` that is too small truncates a target (the launch forces <eos> at the end): the kernel leaves a flag
            # in a word this module keeps - read it at a sync point with target_overflow() (the trainer does after its calibration)
            ov = self.__dict__.get("_overflow")
            if ov is None or ov.device != targets.device:
                ov = self.__dict__["_overflow"] = torch.zeros(1, dtype=torch.int32, device=targets.device)
            return ops.decoder_targets(targets, self.sos_id, self.eos_id, umax, overflow=ov)      # one launch
        comp, n = _compact_targets(targets)
        if umax is None:
            umax = int(n.max().item())
        ys = comp[:, :umax]
        B = targets.shape[0]
        ys_in = torch.cat([torch.full((B, 1), self.sos_id, dtype=targets.dtype, device=targets.device), ys], 1)
        ys_out = torch.cat([ys, torch.zeros((B, 1), dtype=targets.dtype, device=targets.device)], 1)
        ys_out.scatter_(1, n[:, None], self.eos_id)
        return ys_in, ys_out, ((ys_in > 0).sum(1)).to(torch.int32)

    def target_overflow(self):
        """True if some preprocess call since the module was built was handed a `umax` smaller than a target's length (host sync)."""
        ov = self.__dict__.get("_overflow")
        return bool(int(ov)) if ov is not None else False

    def _impl(self, targets, enc, enc_len):
        ys_in, ys_out, dec_len = self._preprocess(targets)
        B, U = ys_in.shape
        dp = _drop(self, "dropout")   # decoder.py:83
        x32, x16 = ops.embed_pe(ys_in, self.tgt_word_emb.weight.detach().float(), self.positional_encoding.rows(U),
                                want_bf16=(_PRECISION == "bf16"), drop=dp)
        x = Act(x32, x16, B, U)
        if _TAPE is not None:
            x_emb, emb = x, self.tgt_word_emb

            def bw_emb():
                ops.embed_bwd(ys_in, x_emb.grad, emb.weight.grad, drop=dp)
                x_emb.grad = None

            _TAPE.push(bw_emb, (emb.weight,))
        cross = self._cross_kv(enc)
        # the decoder's small attention masks: one launch for all self-attention calls, one for all cross-attention calls (queueing
        # them ahead on a side stream like the encoder's measured neutral; 12 launches on this latency-bound chain are 12 x ~5 us.
        # Round 4, under the executor: forked in front of the cross K / V projection so that the 55 us of hashing run beside that
        # 78 us GEMM - 11.18-11.54 against 11.12-11.17 ms per step: the extra chain's events cost more than the overlap gives)
        n = len(self.layer_stack)
        m_self = m_cross = [None] * n
        if self.training and x32.is_cuda and n <= 8:
            ds = [_drop(layer.slf_attn, "attention.dropout") for layer in self.layer_stack]
            dc = [_drop(layer.enc_attn, "attention.dropout") for layer in self.layer_stack]
            if all(d is not None for d in ds + dc):
                m_self = [(d, b_, None) for d, b_ in zip(ds, ops.attention_dropmask_multi(ds, B, self.n_head, U, U, x32.device))]
                m_cross = [(d, b_, None) for d, b_ in zip(dc, ops.attention_dropmask_multi(dc, B, self.n_head, U, enc.L, x32.device))]
            elif any(d is not None for d in ds + dc):
                m_self = [(d, ops.attention_dropmask(d, B, self.n_head, U, U, x32.device), None) for d in ds]
                m_cross = [(d, ops.attention_dropmask(d, B, self.n_head, U, enc.L, x32.device), None) for d in dc]
        for i, layer in enumerate(self.layer_stack):
            x = layer._impl(x, enc, dec_len, enc_len, kv_pre=cross(i), attn_drop=(m_self[i], m_cross[i]))
        logits = _vocab_proj(self, "prj", self.tgt_word_prj.weight, x)
        return logits.view(B, U, self.n_tgt_vocab), ys_out

    @torch.no_grad()
    def step(self, prefixs, encoded, len_encoded):
        """decoder.py:98-120 - log-softmax scores [B, V] of the next token after `prefixs` int64 [B, i] (every prefix position
        counts: the step's masks are causal-only).  The whole prefix is recomputed, like the reference; `batch_decode` does not."""
        B, U = prefixs.shape
        enc = _act(encoded)
        enc_len = ops.as_i32(len_encoded, encoded.device)
        dec_len = torch.full((B,), U, dtype=torch.int32, device=encoded.device)
        x32, x16 = ops.embed_pe(prefixs.contiguous(), self.tgt_word_emb.weight.detach().float(), self.positional_encoding.rows(U),
                                want_bf16=(_PRECISION == "bf16"))
        x = Act(x32, x16, B, U)
        cross = self._cross_kv(enc)
        for i, layer in enumerate(self.layer_stack):
            x = layer._impl(x, enc, dec_len, enc_len, kv_pre=cross(i))
        last = Act(x.f32.view(B, U, -1)[:, -1].contiguous(), None if x.b16 is None else x.b16.view(B, U, -1)[:, -1].contiguous(), B, 1)
        return ops.log_softmax_rows(_vocab_proj(self, "prj", self.tgt_word_prj.weight, last))

    @torch.no_grad()
    def batch_decode(self, encoded, len_encoded, max_decode_len=100):
        """decoder.py:138-164 - greedy decoding of a batch: -> (preds int64 [B, steps], len_decoded, torch.zeros(0)).
        One new token per step through the layers, against per-layer self-attention K/V caches and the encoder-side K/V of all
        layers projected once.  The step keeps everything that changes from token to token in DEVICE memory (position, cache
        slot, key length, finished flags: asr_decode_embed / asr_kv_cache_put / asr_decode_advance), so its ~70 launches are
        captured once as a hipGraph and replayed per token; the reference's per-step host check `finished.all()` becomes one
        4-byte read every 8 tokens (the device stops advancing by itself once every row has produced <eos>)."""
        B, L = encoded.shape[0], encoded.shape[1]
        dev = encoded.device
        T = int(max_decode_len)
        if T <= 0:
            return torch.zeros((B, 0), dtype=torch.long, device=dev), torch.zeros_like(len_encoded), torch.zeros(0)
        key = (B, L, T, str(dev), _PRECISION, _PARAM_EPOCH, self.sos_id, self.eos_id, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        dg = self.__dict__.get("_decode_graph")
        if dg is None or dg["key"] != key:
            dg = self._build_decode_graph(B, L, T, dev, key)
            self.__dict__["_decode_graph"] = dg
        dg["enc"].copy_(encoded.reshape(B * L, -1))
        dg["enc_len"].copy_(ops.as_i32(len_encoded, dev))
        dg["state"].copy_(dg["state0"])
        dg["k_len"].fill_(1)
        dg["finished"].zero_()
        dg["len_decoded"].fill_(1)
        dg["preds"].zero_()
        dg["preds"][:, 0] = self.sos_id
        dg["cur"].fill_(self.sos_id)
        dg["prologue"]()
        steps = T
        for t in range(T):
            dg["step"]()
            if (t & 7) == 7 or t == T - 1:
                stop = int(dg["state"][1])
                if stop >= 0:
                    steps = stop
                    break
        fin = dg["finished"].to(len_encoded.dtype)
        len_decoded = dg["len_decoded"].to(len_encoded.dtype) - (1 - fin)      # for decoded length cut by encoded length (decoder.py:161)
        return dg["preds"][:, 1:steps + 1].clone(), len_decoded, torch.zeros(0)

    @torch.no_grad()
    def batch_beam_decode(self, encoded, len_encoded, beam_size=1, max_decode_len=100):
        """decoder.py:166-234 - beam search over a batch -> (preds int64 [B, beam, steps], len_decoded [B, beam], scores f32 [B, beam]),
        beams sorted by score.  The reference's arithmetic and quirks - initial scores [0, -1e10, ...] per utterance, log_softmax
        applied to `step`'s log-probabilities again (:191), `finished` / `len_decoded` staying with the beam SLOT when the beams are
        re-gathered - on the decode path's machinery instead of a per-step recompute of the whole prefix: the B * beam hypotheses are
        rows that feed ONE new position per step through the layers against per-layer self-attention K / V caches (re-gathered by
        parent row after the pruning); the beams of an utterance are `beam` query positions of ONE cross-attention over that
        utterance's encoder K / V (projected once for all layers, not replicated per beam); position, stop flag and <eos> bookkeeping
        live in device memory (asr_beam_step / asr_beam_reorder_cache / asr_beam_advance), so the step is captured once per shape and
        replayed; the reference's per-step `finished.all()` becomes a 4-byte read every 8 steps."""
        B, L = encoded.shape[0], encoded.shape[1]
        beam, T = int(beam_size), int(max_decode_len)
        dev = encoded.device
        if T <= 0:
            z = torch.zeros((B, beam), dtype=len_encoded.dtype, device=dev)
            return (torch.zeros((B, beam, 0), dtype=torch.long, device=dev), z,
                    torch.tensor([0.0] + [-1e10] * (beam - 1), dtype=torch.float32, device=dev).repeat(B).view(B, beam))
        key = (B, L, T, beam, str(dev), _PRECISION, _PARAM_EPOCH, self.sos_id, self.eos_id, tuple((p.data_ptr(), p._version) for p in self.parameters()))
        g = self.__dict__.get("_beam_graph")
        if g is None or g["key"] != key:
            self.__dict__["_beam_graph"] = None
            g = self._build_beam_graph(B, L, T, beam, dev, key)
            self.__dict__["_beam_graph"] = g
        g["enc"].copy_(encoded.reshape(B * L, -1))
        g["enc_len"].copy_(ops.as_i32(len_encoded, dev))
        g["state"].copy_(g["state0"])
        g["k_len"].fill_(1)
        g["finished"].zero_()
        g["len_decoded"].fill_(1)
        g["preds"].fill_(self.sos_id)
        g["cur"].fill_(self.sos_id)
        g["scores"].copy_(g["scores0"])
        g["prologue"]()
        steps = T
        for t in range(T):
            g["step"]()
            if (t & 7) == 7 or t == T - 1:
                stop = int(g["state"][1])
                if stop >= 0:
                    steps = stop
                    break
        fin = g["finished"].to(len_encoded.dtype)
        len_decoded = g["len_decoded"].to(len_encoded.dtype) - (1 - fin)
        scores_sorted, order = ops.topk_rows(g["scores"].view(B, beam), beam)
        order = (torch.arange(B, device=dev)[:, None] * beam + order).view(-1)
        return g["preds"][:, 1:steps + 1][order].view(B, beam, -1), len_decoded[order].view(B, beam), scores_sorted

    def _build_beam_graph(self, B, L, T, beam, dev, key):
        n, h, cdt, d, N = len(self.layer_stack), self.n_head, _cdtype(), self.d_model, B * beam
        enc_buf = torch.zeros((B * L, d), device=dev, dtype=torch.float32)
        enc_len = torch.ones(B, dtype=torch.int32, device=dev)
        state0 = torch.tensor([0, -1], dtype=torch.int32, device=dev)
        state = state0.clone()
        k_len = torch.ones(N, dtype=torch.int32, device=dev)
        finished = torch.zeros(N, dtype=torch.uint8, device=dev)
        len_decoded = torch.ones(N, dtype=torch.int64, device=dev)
        preds = torch.full((N, T + 1), self.sos_id, dtype=torch.long, device=dev)
        scores0 = torch.tensor([0.0] + [-1e10] * (beam - 1), dtype=torch.float32, device=dev).repeat(B)
        scores = scores0.clone()
        cur = torch.full((N,), self.sos_id, dtype=torch.long, device=dev)
        parent = torch.zeros(N, dtype=torch.long, device=dev)
        n_steps = torch.full((B,), T + 1, dtype=torch.int32, device=dev)          # every utterance takes every step (the <eos> stop is global)
        cache = torch.zeros((2 * n, N, h, T, 64), device=dev, dtype=cdt)
        emb = self.tgt_word_emb.weight.detach().float()
        pe = self.positional_encoding.pe[0].contiguous()
        box = {}

        def prologue():
            enc = Act(enc_buf, None, B, L)
            box["enc"], box["cross"] = enc, self._cross_kv(enc)

        def step():
            with _decode_step():
                _step_body()

        def _step_body():
            x32, x16 = ops.decode_embed(cur, emb, pe, state, want_bf16=(_PRECISION == "bf16"))
            x = Act(x32, x16, N, 1)
            for i, layer in enumerate(self.layer_stack):
                x = layer.slf_attn._impl_cached_self(x, cache[2 * i], cache[2 * i + 1], state, k_len, next_attn=layer.enc_attn, next_lq=beam)
                xq = Act(x.f32, x.b16, B, beam)               # the beams of an utterance: `beam` queries of one cross-attention
                x = layer._decode_cross_ffn(xq, box["enc"], enc_len, box["cross"](i), N, q=x.next_q)
            # Decoder.step's log-probabilities, log_softmax again (decoder.py:118 + :191), top-k: one launch
            best, ids = ops.lsm_topk_rows(_vocab_proj(self, "prj", self.tgt_word_prj.weight, x), beam, twice=True)
            ops.beam_step(scores, best, ids, preds, state, n_steps, parent, cur, beam)
            ops.beam_reorder_cache(cache, parent, state, beam)
            ops.beam_advance(state, k_len, cur, self.eos_id, finished, len_decoded)

        g = _DecodeGraph(key=key, enc=enc_buf, enc_len=enc_len, state=state, state0=state0, k_len=k_len, finished=finished,
                         len_decoded=len_decoded, preds=preds, scores=scores, scores0=scores0, cur=cur)
        g.keep = (cache, parent, n_steps, emb, pe, box, prologue, step)     # the captured kernels address these by raw pointer
        prologue()                      # eager warm-up of both parts: code objects, derived weights, allocator pools
        step()
        torch.cuda.synchronize(dev)
        if os.environ.get("ASR_AMD_DECODE_GRAPH", "1") != "0":
            try:
                gp, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp):
                    prologue()
                with torch.cuda.graph(gs, pool=gp.pool()):
                    step()
                g["prologue"], g["step"], g["graphs"] = gp.replay, gs.replay, (gp, gs)
                return g
            except Exception as e:          # not capturable on this stack: search eagerly, remember why
                import warnings
                warnings.warn("asr_amd.Decoder: hipGraph capture of the beam step failed, searching eagerly (%s: %s)" % (type(e).__name__, e))
                torch.cuda.synchronize(dev)
        g["prologue"], g["step"], g["graphs"] = prologue, step, None
        return g

    def _build_decode_graph(self, B, L, T, dev, key):
        n, h = len(self.layer_stack), self.n_head
        cdt = _cdtype()
        d = self.d_model
        enc_buf = torch.zeros((B * L, d), device=dev, dtype=torch.float32)
        enc_len = torch.ones(B, dtype=torch.int32, device=dev)
        state = torch.zeros(2, dtype=torch.int32, device=dev)
        state0 = torch.tensor([0, -1], dtype=torch.int32, device=dev)
        k_len = torch.ones(B, dtype=torch.int32, device=dev)
        finished = torch.zeros(B, dtype=torch.uint8, device=dev)
        len_decoded = torch.ones(B, dtype=torch.int64, device=dev)
        preds = torch.zeros((B, T + 1), dtype=torch.long, device=dev)
        cur = torch.zeros(B, dtype=torch.long, device=dev)
        kc = [torch.zeros((B, h, T, 64), device=dev, dtype=cdt) for _ in range(n)]
        vc = [torch.zeros((B, h, T, 64), device=dev, dtype=cdt) for _ in range(n)]
        emb = self.tgt_word_emb.weight.detach().float()
        pe = self.positional_encoding.pe[0].contiguous()
        box = {}

        def prologue():
            enc = Act(enc_buf, None, B, L)
            box["enc"], box["cross"] = enc, self._cross_kv(enc)

        def step():
            with _decode_step():
                _step_body()

        def _step_body():
            x32, x16 = ops.decode_embed(cur, emb, pe, state, want_bf16=(_PRECISION == "bf16"))
            x = Act(x32, x16, B, 1)
            for i, layer in enumerate(self.layer_stack):
                x = layer.slf_attn._impl_cached_self(x, kc[i], vc[i], state, k_len, next_attn=layer.enc_attn, next_lq=1)
                x = layer._decode_cross_ffn(x, box["enc"], enc_len, box["cross"](i), B, q=x.next_q)
            ops.argmax_rows(_vocab_proj(self, "prj", self.tgt_word_prj.weight, x), out=cur)      # argmax of log_softmax = argmax of the logits
            ops.decode_advance(cur, preds, state, k_len, finished, len_decoded, self.eos_id)

        g = _DecodeGraph(key=key, enc=enc_buf, enc_len=enc_len, state=state, state0=state0, k_len=k_len, finished=finished,
                         len_decoded=len_decoded, preds=preds, cur=cur)
        # everything the captured kernels address must outlive the capture: the graphs hold raw pointers, not references (the K/V
        # caches live in the ordinary allocator pool - dropped with these closures they would be handed to the next torch.empty)
        g.keep = (kc, vc, emb, pe, box, prologue, step)
        state.copy_(state0)
        use_graph = os.environ.get("ASR_AMD_DECODE_GRAPH", "1") != "0"
        prologue()                      # eager warm-up of both parts: code objects, derived weights, allocator pools
        step()
        torch.cuda.synchronize(dev)
        if use_graph:
            try:
                gp, gs = torch.cuda.CUDAGraph(), torch.cuda.CUDAGraph()
                with torch.cuda.graph(gp):
                    prologue()
                with torch.cuda.graph(gs, pool=gp.pool()):
                    step()
                g["prologue"], g["step"], g["graphs"] = gp.replay, gs.replay, (gp, gs)
                return g
            except Exception as e:          # not capturable on this stack: decode eagerly, remember why
                import warnings
                warnings.warn("asr_amd.Decoder: hipGraph capture of the decode step failed, decoding eagerly (%s: %s)" % (type(e).__name__, e))
                torch.cuda.synchronize(dev)
        g["prologue"], g["step"], g["graphs"] = prologue, step, None
        return g

    def cross_kv_params(self):
        """([w_ks, w_vs weights of layer 0, 1, ...], [their biases]) - the trainer keeps each list adjacent in its flat buffers."""
        att = [layer.enc_attn for layer in self.layer_stack]
        return ([w for a in att for w in (a.w_ks.weight, a.w_vs.weight)], [b for a in att for b in (a.w_ks.bias, a.w_vs.bias)])

    def _cross_kv(self, enc):
        """The encoder-side keys / values of every decoder layer depend only on the encoder output: project them with ONE
        [B*L, n_layers*2*h*64] GEMM instead of n_layers narrow ones, and in the backward collect every layer's dK / dV in one
        [B*L, n_layers*2*h*64] buffer so the weight gradient and the gradient wrt the encoder output are one GEMM each (K = 3072
        instead of six K = 512 passes that each re-read and re-write the [B*L, d_model] accumulator)."""
        ws, bs = self.cross_kv_params()
        n, h = len(self.layer_stack), self.n_head
        hd = h * 64
        W, bias = self._w("cross_kv", tuple(ws)), self._b("cross_kvb", tuple(bs))
        kv = ops.proj_heads(enc.mma(), W, bias, 2 * n, enc.B, enc.L, h, 1.0)
        if _TAPE is None:
            return lambda i: (kv[2 * i], kv[2 * i + 1], None)
        box = {}

        def dkv_of(i):
            def get():
                if "dkv" not in box:
                    box["dkv"] = torch.empty((enc.B * enc.L, 2 * n * hd), device=kv.device, dtype=_cdtype())
                d = box["dkv"]
                return d[:, 2 * i * hd:(2 * i + 1) * hd], d[:, (2 * i + 1) * hd:(2 * i + 2) * hd]
            return get

        def bw():   # pushed before the layers -> runs after all of them have written their dK / dV columns
            dkv = box.pop("dkv")
            _wg(dkv, enc.mma(), out=_gcat(ws), accumulate=True, colsum=_gcat(bs))
            if enc.lazy_grad is not None:
                enc.lazy_grad()      # (the CTC branch's gradient: it becomes this GEMM's addend)
            enc.grad = ops.gemm_nn(dkv, W, addend=enc.grad)

        _TAPE.push(bw, tuple(ws) + tuple(bs))
        return lambda i: (kv[2 * i], kv[2 * i + 1], dkv_of(i))

    def forward(self, targets, encoder_padded_outputs, encoder_input_lengths):
        enc = _act(encoder_padded_outputs)
        return self._impl(targets, enc, ops.as_i32(encoder_input_lengths, encoder_padded_outputs.device))


class Decoder_CIF(_Cached):
    """src/transformer/decoder.py:327-399 (forward path)."""

    def __init__(self, sos_id, n_tgt_vocab, n_layers, n_head, d_model, d_inner, dropout=0.1):
        super().__init__()
        self.sos_id, self.n_tgt_vocab, self.d_word_vec = sos_id, n_tgt_vocab, d_model
        self.n_layers, self.n_head, self.d_model, self.d_inner = n_layers, n_head, d_model, d_inner
        self.d_output, self.dropout_rate = n_tgt_vocab, dropout
        self.tgt_word_emb = nn.Embedding(n_tgt_vocab, d_model)
        self.positional_encoding = PositionalEncoding(d_model)
        self.layer_stack = nn.ModuleList([EncoderLayer(d_model, d_inner, n_head, dropout=dropout) for _ in range(n_layers)])
        self.input_affine = nn.Linear(2 * d_model, d_model, bias=False)
        self.tgt_word_prj = nn.Linear(2 * d_model, n_tgt_vocab, bias=False)
        nn.init.xavier_normal_(self.tgt_word_prj.weight)

    def preprocess(self, target):
        pad_mask = (target > 0).long()
        sos = torch.full((target.size(0), 1), self.sos_id, dtype=torch.long, device=target.device)
        return torch.cat([sos, target[:, :-1]], 1) * pad_mask

    def forward(self, encoded_attentioned, target, cif_slot=None):
        """cif_slot: {"g": tensor or None} - when the tape is recording, the gradient wrt `encoded_attentioned` is accumulated there."""
        B, U, D = encoded_attentioned.shape
        rec = _TAPE is not None and cif_slot is not None
        if target.is_cuda and target.dtype == torch.int64:
            ys_in, dec_len = ops.decoder_cif_targets(target, self.sos_id)      # preprocess + lengths, one launch
        else:
            ys_in = self.preprocess(target)
            dec_len = (target > 0).sum(1).to(torch.int32)          # tail padding (every reference data path)
        cif32 = encoded_attentioned.contiguous().float().view(B * U, D)
        dp = _drop(self, "dropout")   # decoder.py:385
        e32, _ = ops.embed_pe(ys_in, self.tgt_word_emb.weight.detach().float(), self.positional_encoding.rows(U), drop=dp)
        cat1 = torch.cat([cif32, e32], -1)
        w_in = self._w("inaff", (self.input_affine.weight,))
        a = Act(ops.gemm_nt(cat1, w_in, None), None, B, U)
        if rec:
            a0, emb, aff = a, self.tgt_word_emb, self.input_affine

            def bw_in():
                ops.gemm_tn(a0.grad, cat1, out=aff.weight.grad, accumulate=True)
                d_cat = ops.gemm_nn(a0.grad, w_in)
                a0.grad = None
                cif_slot["g"] = d_cat[:, :D].contiguous() if cif_slot["g"] is None else _add_into(cif_slot["g"], d_cat[:, :D])
                ops.embed_bwd(ys_in, d_cat[:, D:].contiguous(), emb.weight.grad, drop=dp)

            _TAPE.push(bw_in, (aff.weight, emb.weight))
        for layer in self.layer_stack:
            a = layer._impl(a, dec_len, causal=True)
        cat2 = Act(torch.cat([cif32, a.f32], -1), None, B, U)
        if rec:
            a_last = a

            def bw_split():   # runs after the vocab projection's closure has produced cat2.grad
                g = cat2.grad
                cat2.grad = None
                cif_slot["g"] = g[:, :D].contiguous() if cif_slot["g"] is None else _add_into(cif_slot["g"], g[:, :D])
                _acc(a_last, g[:, D:].contiguous())

            _TAPE.push(bw_split, ())
        logits = _vocab_proj(self, "prj", self.tgt_word_prj.weight, cat2)
        return logits.view(B, U, self.n_tgt_vocab)

    # ---- inference (decoder.py:401-552) -----------------------------------------------------------------------------------------
    def _prefix_layers(self, ys, frames32):
        """input_affine([frames | emb(ys) + pe]) through the layers under the causal mask -> Act [N * U, d] (no pad mask: every
        prefix position counts, decoder.py:403-404)"""
        N, U = ys.shape
        e32, _ = ops.embed_pe(ys.contiguous(), self.tgt_word_emb.weight.detach().float(), self.positional_encoding.rows(U))
        a = Act(ops.gemm_nt(torch.cat([frames32, e32], -1), self._w("inaff", (self.input_affine.weight,)), None), None, N, U)
        return a

    def _last_scores(self, frame_last32, x_last32):
        N = x_last32.shape[0]
        cat2 = Act(torch.cat([frame_last32, x_last32], -1).contiguous(), None, N, 1)
        return ops.log_softmax_rows(_vocab_proj(self, "prj", self.tgt_word_prj.weight, cat2))

    @torch.no_grad()
    def step_forward(self, ys, encoded_attentioned, t):
        """decoder.py:401-423 - log-softmax scores [N, V] of the token after the prefix `ys` int64 [N, t + 1]; the whole prefix is
        recomputed, like the reference (`recognize_beam` below does not)."""
        N, U = ys.shape
        D = encoded_attentioned.shape[-1]
        frames = encoded_attentioned[:, :t + 1].contiguous().float()
        a = self._prefix_layers(ys, frames.view(N * U, D))
        dec_len = torch.full((N,), U, dtype=torch.int32, device=ys.device)
        for layer in self.layer_stack:
            a = layer._impl(a, dec_len, causal=True)
        return self._last_scores(frames[:, -1], a.f32.view(N, U, -1)[:, -1])

    @torch.no_grad()
    def step_forward_cache(self, ys, enc_attentioned, dec_cache, t):
        """decoder.py:477-496 - the reference's cached step, same contract: `dec_cache` [N, t, n_layers, d] holds every layer's
        outputs at the earlier positions -> (scores [N, V], new cache [N, t + 1, n_layers, d]).  Each layer is evaluated on its whole
        input under the causal mask and the last row kept (= EncoderLayer.forward_cache, encoder.py:81-87: the last query against all
        positions); `recognize_beam` keeps per-layer K / V instead and never re-reads earlier positions."""
        N, U = ys.shape
        D = enc_attentioned.shape[-1]
        frames = enc_attentioned[:, :t + 1].contiguous().float()
        x = self._prefix_layers(ys, frames.view(N * U, D))
        dec_len = torch.full((N,), U, dtype=torch.int32, device=ys.device)
        new_cache = []
        for i, layer in enumerate(self.layer_stack):
            last = layer._impl(x, dec_len, causal=True).f32.view(N, U, -1)[:, -1:]
            full = torch.cat([dec_cache[:, :, i].to(last.dtype), last], 1).contiguous()
            new_cache.append(full.unsqueeze(2))
            x = Act(full.view(N * U, -1), None, N, U)
        return self._last_scores(frames[:, -1], x.f32.view(N, U, -1)[:, -1]), torch.cat(new_cache, 2)

    @torch.no_grad()
    def recognize_beam(self, encoded_attentioned, char_list, args):
        """decoder.py:425-475 - beam search over ONE utterance's integrated frames [1, U, d]: exactly U steps (no <eos> handling),
        each live hypothesis extended by its `beam` best tokens, candidates (hypothesis-major, rank-minor) cut to the `beam` best by
        accumulated score with ties in candidate order (Python's stable sort).  -> ([token lists incl. <sos>], [their lengths]), the
        `nbest` best.  Runs `batch_recognize_beam` on a batch of one."""
        U = int(encoded_attentioned.shape[1])
        n = torch.full((1,), U, dtype=torch.int32, device=encoded_attentioned.device)
        return self.batch_recognize_beam(encoded_attentioned[:1], n, args.beam_size, args.nbest, n_host=[U])[0]

    @torch.no_grad()
    def batch_recognize_beam(self, frames, n_frames, beam_size, nbest=1, n_host=None):
        """`recognize_beam` for B utterances at once: frames f32 [B, Umax, d], utterance b decodes exactly n_frames[b] steps ->
        [(token lists incl. <sos>, lengths)] per utterance, each what the reference's per-utterance search returns.
        MI355X form: the B * beam hypotheses are the rows of one batch; per step ONE new position goes through the layers against
        per-layer K / V caches, the pruning (asr_topk_rows, asr_beam_step) gathers token rows and scores in place and the caches are
        re-gathered by parent row (asr_beam_reorder_cache); the position lives in device memory, so the ~60 launches of a step are
        captured once per (B, beam, length bucket) as a hipGraph and replayed.  The first step runs `beam` copies of <sos> per
        utterance with scores [0, -1e10, ...], so only the first copy's extensions survive (needs beam <= vocabulary size)."""
        beam, nbest = int(beam_size), int(nbest)
        B, Umax, D = int(frames.shape[0]), int(frames.shape[1]), int(frames.shape[2])
        dev = frames.device
        if beam > self.n_tgt_vocab:
            raise ValueError("recognize_beam: beam_size %d exceeds the vocabulary (%d)" % (beam, self.n_tgt_vocab))
        if n_host is None:
            n_host = [int(v) for v in n_frames.tolist()]
        steps = max(n_host) if n_host else 0
        if steps > Umax:
            raise ValueError("recognize_beam: %d steps asked of %d integrated frames" % (steps, Umax))
        if steps > 0:
            Tmax = max(32, (steps + 31) // 32 * 32)
            key = (B, beam, Tmax, D, str(dev), _PRECISION, _PARAM_EPOCH, self.sos_id, tuple((p.data_ptr(), p._version) for p in self.parameters()))
            g = self.__dict__.get("_beam_graph")
            if g is None or g["key"] != key:
                self.__dict__["_beam_graph"] = None
                g = self._build_beam_graph(B, beam, Tmax, D, dev, key)
                self.__dict__["_beam_graph"] = g
            g["frames"][:, :steps].copy_(frames[:, :steps])
            g["n_steps"].copy_(ops.as_i32(n_frames, dev))
            g["state"].copy_(g["state0"])
            g["k_len"].fill_(1)
            g["preds"].fill_(self.sos_id)
            g["cur"].fill_(self.sos_id)
            g["scores"].copy_(g["scores0"])
            for _ in range(steps):
                g["step"]()
            preds = g["preds"].view(B, beam, -1).cpu()
        out = []
        for b, n in enumerate(n_host):
            if n == 0:
                out.append(([[self.sos_id]], [1]))
                continue
            ys = preds[b, :min(beam, nbest), :n + 1].tolist()
            out.append((ys, [len(y) for y in ys]))
        return out

    def _build_beam_graph(self, B, beam, Tmax, D, dev, key):
        n, h, cdt, N = len(self.layer_stack), self.n_head, _cdtype(), B * beam
        frames = torch.zeros((B, Tmax, D), device=dev, dtype=torch.float32)
        n_steps = torch.zeros(B, dtype=torch.int32, device=dev)
        state0 = torch.tensor([0, -1], dtype=torch.int32, device=dev)
        state = state0.clone()
        k_len = torch.ones(N, dtype=torch.int32, device=dev)
        cache = torch.zeros((2 * n, N, h, Tmax, 64), device=dev, dtype=cdt)        # [K of layer 0, V of layer 0, K of layer 1, ...]
        preds = torch.full((N, Tmax + 1), self.sos_id, dtype=torch.long, device=dev)
        scores0 = torch.tensor([0.0] + [-1e10] * (beam - 1), dtype=torch.float32, device=dev).repeat(B)
        scores = scores0.clone()
        cur = torch.full((N,), self.sos_id, dtype=torch.long, device=dev)
        parent = torch.zeros(N, dtype=torch.long, device=dev)
        emb = self.tgt_word_emb.weight.detach().float()
        pe = self.positional_encoding.pe[0].contiguous()

        def step():
            with _decode_step():
                _step_body()

        def _step_body():
            cat1 = ops.beam_cat_frames(frames, state, beam, cur=cur, emb=emb, pe=pe)            # decoder.py:407-408
            x = Act(ops.gemm_nt(cat1, self._w("inaff", (self.input_affine.weight,)), None), None, N, 1)
            for j, layer in enumerate(self.layer_stack):
                x = layer.slf_attn._impl_cached_self(x, cache[2 * j], cache[2 * j + 1], state, k_len)
                x = layer.pos_ffn._impl(x, None)
            cat2 = Act(ops.beam_cat_frames(frames, state, beam, other=x.f32.contiguous()), None, N, 1)     # decoder.py:416
            best, ids = ops.lsm_topk_rows(_vocab_proj(self, "prj", self.tgt_word_prj.weight, cat2), beam)
            ops.beam_step(scores, best, ids, preds, state, n_steps, parent, cur, beam)
            ops.beam_reorder_cache(cache, parent, state, beam)
            ops.beam_advance(state, k_len)

        g = _DecodeGraph(key=key, frames=frames, n_steps=n_steps, state=state, state0=state0, k_len=k_len, preds=preds, scores=scores,
                         scores0=scores0, cur=cur)
        g.keep = (cache, parent, emb, pe, step)     # the captured kernels address these by raw pointer
        n_steps.fill_(1)
        step()                          # eager warm-up: code objects, derived weights, allocator pools
        torch.cuda.synchronize(dev)
        if os.environ.get("ASR_AMD_DECODE_GRAPH", "1") != "0":
            try:
                gs = torch.cuda.CUDAGraph()
                with torch.cuda.graph(gs):
                    step()
                g["step"], g["graphs"] = gs.replay, (gs,)
                return g
            except Exception as e:          # not capturable on this stack: search eagerly, remember why
                import warnings
                warnings.warn("asr_amd.Decoder_CIF: hipGraph capture of the beam step failed, searching eagerly (%s: %s)" % (type(e).__name__, e))
                torch.cuda.synchronize(dev)
        g["step"], g["graphs"] = step, None
        return g

    recognize_beam_cache = recognize_beam      # decoder.py:498-552: the cached search IS the implementation above


# ------------------------------------------------------------------------------------------------------------
class _TapeFn(torch.autograd.Function):
    """Bridges torch.autograd to the HIP backward tape: forward records the tape; backward seeds each differentiable output's
    gradient slot, replays the tape (HIP kernels write into a flat scratch gradient buffer laid out like the trainer's) and
    returns one gradient per parameter, so `loss.backward()` + any torch optimizer work exactly as with the reference's modules."""

    @staticmethod
    def forward(ctx, model, run, holder, *params):
        with torch.no_grad(), record() as tape:
            outs, slots, extra = run()
        holder["extra"] = extra
        ctx.model, ctx.tape, ctx.params, ctx.slots = model, tape, params, slots
        return tuple(outs)

    @staticmethod
    def backward(ctx, *gouts):
        from .trainer import _param_order, flat_offsets
        model, tape, params = ctx.model, ctx.tape, ctx.params
        order = _param_order(model)
        off_list, n = flat_offsets(order)
        offs = {id(p): o for p, o in zip(order, off_list)}
        dev = order[0].device
        scratch = torch.zeros((n + 63) // 64 * 64, device=dev, dtype=torch.float32)
        saved = []
        for p in order:
            saved.append((p, p.grad, getattr(p, "_asr_off", None), getattr(p, "_asr_gflat", None)))
            off = offs[id(p)]
            p.grad = scratch[off:off + p.numel()].view(p.shape)
            p._asr_off, p._asr_gflat = off, scratch
        try:
            with torch.no_grad():
                for g, slot in zip(gouts, ctx.slots):
                    slot["g"] = g if g is not None else torch.zeros(slot["shape"], device=dev)
                tape.backward()
            grads = tuple(p.grad for p in params)
        finally:
            for p, g, off, gf in saved:
                p.grad = g
                if off is None:
                    del p._asr_off, p._asr_gflat
                else:
                    p._asr_off, p._asr_gflat = off, gf
        return (None, None, None) + grads


def _taped(model, run):
    """run() -> (differentiable outputs, their gradient slots, extra).  Attaches the tape-backed autograd node when wanted."""
    if not _autograd_wanted(model):
        outs, _, extra = run()
        return list(outs), extra
    holder = {}
    params = [p for p in model.parameters() if p.requires_grad]
    outs = _TapeFn.apply(model, run, holder, *params)
    return list(outs), holder["extra"]


def _slots(*mods_keys):
    """gradient slots registered by _vocab_proj while recording: _slots(modA, "keyA", modB, "keyB") (None when not recording)"""
    return [mods_keys[i].__dict__.get("_grad_slots", {}).get(mods_keys[i + 1]) for i in range(0, len(mods_keys), 2)]


def _autograd_wanted(model):
    return (torch.is_grad_enabled() and _TAPE is None and _PRECISION == "bf16" and
            any(p.requires_grad for p in model.parameters()))


def _xavier_all(model):
    for p in model.parameters():
        if p.dim() > 1:
            nn.init.xavier_uniform_(p)


class Transformer(_Cached):
    """src/transformer/transformer.py:7-35."""

    def __init__(self, encoder, decoder, spec_aug_cfg=None):
        super().__init__()
        self.encoder, self.decoder, self.spec_aug_cfg = encoder, decoder, spec_aug_cfg
        _xavier_all(self)

    def _augment(self, features, len_features):
        """transformer.py:28-29 / :116-117 / :142-143, cif_model.py:31-32: SpecAugment whenever a config is set (in place on the batch,
        like the reference), on the device (data.spec_aug)."""
        if self.spec_aug_cfg:
            from .data import spec_aug
            features, len_features = spec_aug(features, len_features, self.spec_aug_cfg)
        return features, len_features

    def forward(self, features, len_features, padded_target):
        _assign_names(self)
        features, len_features = self._augment(features, len_features)
        def run():
            lens = ops.as_i32(len_features, features.device)
            enc = self.encoder._impl(_act(features), lens)
            logits, targets_eos = self.decoder._impl(padded_target, enc, lens)
            return [logits], [self.decoder.__dict__.get("_grad_slots", {}).get("prj")], targets_eos
        (logits,), targets_eos = _taped(self, run)
        return logits, targets_eos


class CTC_Transformer(Transformer):
    """src/transformer/transformer.py:100-124 — returns (ctc_len, ctc_logits, (logits, targets_eos))."""

    def __init__(self, encoder, decoder, spec_aug_cfg=None):
        super().__init__(encoder, decoder, spec_aug_cfg)
        self.ctc_fc = nn.Linear(encoder.d_output, decoder.d_output, bias=False)

    def _ctc_logits(self, enc, lens=None):
        hook = self.__dict__.get("_ctc_hook")      # trainer: the whole CTC branch (projection, loss, both backwards) on a side stream
        if hook is not None and _TAPE is not None:
            return hook(enc, lens)
        return _vocab_proj(self, "ctc", self.ctc_fc.weight, enc)

    def forward(self, features, len_features, padded_target):
        _assign_names(self)
        features, len_features = self._augment(features, len_features)
        B, L = features.shape[0], features.shape[1]

        def run():
            lens = ops.as_i32(len_features, features.device)
            enc = self.encoder._impl(_act(features), lens)
            ctc_pred = self._ctc_logits(enc, lens)
            logits, targets_eos = self.decoder._impl(padded_target, enc, lens)
            return [ctc_pred, logits], _slots(self, "ctc", self.decoder, "prj"), targets_eos
        (ctc_pred, logits), targets_eos = _taped(self, run)
        return len_features, ctc_pred.view(B, L, -1), (logits, targets_eos)


class Conv_CTC_Transformer(CTC_Transformer):
    """src/transformer/transformer.py:127-153 — returns (ctc_logits, len, logits, targets_eos)."""

    def __init__(self, conv_encoder, encoder, decoder, spec_aug_cfg=None):
        super().__init__(encoder, decoder, spec_aug_cfg)
        self.conv_encoder = conv_encoder

    def forward(self, features, len_features, targets, spec_aug_cfg=False):
        _assign_names(self)
        features, len_features = self._augment(features, len_features)
        def run():
            conv, len_sequence = self.conv_encoder._impl(features, len_features)
            enc = self.encoder._impl(conv, len_sequence)
            ctc_logits = self._ctc_logits(enc, len_sequence)
            logits, targets_eos = self.decoder._impl(targets, enc, len_sequence)
            return [ctc_logits, logits], _slots(self, "ctc", self.decoder, "prj"), (targets_eos, len_sequence, enc.B, enc.L)
        (ctc_logits, logits), (targets_eos, len_sequence, B, L) = _taped(self, run)
        return ctc_logits.view(B, L, -1), len_sequence, logits, targets_eos

    @torch.no_grad()
    def batch_recognize(self, features, len_features, beam_size):
        """transformer.py:172-185 - greedy batch decoding.  As in the reference, the third argument is handed to
        Decoder.batch_decode positionally, where it is `max_decode_len`."""
        conv, len_sequences = self.conv_encoder._impl(features, len_features)
        enc = self.encoder._impl(conv, len_sequences)
        return self.decoder.batch_decode(enc.view3(), len_sequences, beam_size)

    @classmethod
    def create_model(cls, args):
        """transformer.py:188-214."""
        conv_encoder = Conv2dSubsample(d_input=args.d_input * args.LFR_m, d_model=args.d_model, n_layers=args.n_conv_layers)
        encoder = Encoder(d_input=args.d_model, n_layers=args.n_layers_enc, n_head=args.n_head, d_model=args.d_model,
                          d_inner=args.d_inner, dropout=args.dropout)
        decoder = Decoder(sos_id=args.sos_id, eos_id=args.eos_id, n_tgt_vocab=args.vocab_size, n_layers=args.n_layers_dec,
                          n_head=args.n_head, d_model=args.d_model, d_inner=args.d_inner, dropout=args.dropout)
        return cls(conv_encoder, encoder, decoder, spec_aug_cfg=args.spec_aug_cfg)


def cif_forward(hidden, alphas, threshold, max_label_len=None):
    """cif_model.py:57-106 on cif.hip.  Returns (out [B,Umax,H], (fire_idx, n_fire, n_label)).
    One host sync (the reference has the same one: `.max()` used as a tensor size) unless max_label_len is given."""
    cur, rem, fire_idx, n_fire, n_label = ops.cif_scan(alphas.float(), threshold)
    if max_label_len is None:
        stats = torch.stack([n_label.max(), n_fire.max()]).tolist()
        max_label_len = stats[0]
        if stats[1] > max_label_len:
            raise RuntimeError("cif: a row fires %d times but max round(sum alpha) is %d "
                               "(the reference fails here too, cif_model.py:100)" % (stats[1], max_label_len))
    out = ops.cif_gather(hidden.float(), cur, rem, fire_idx, n_fire, int(max_label_len))
    return out, (fire_idx, n_fire, n_label)


class CIF_Model(_Cached):
    """src/transformer/cif_model.py:8-106 — returns (ctc_logits, len, _num, num, logits)."""
    draws_noise = True      # forward() draws torch.rand(B) when no noise is passed (cif_model.py:47): Trainer.step_graphed feeds it in from outside the graph

    def __init__(self, conv_encoder, encoder, assigner, decoder, spec_aug_cfg=None):
        super().__init__()
        self.conv_encoder, self.encoder, self.assigner, self.decoder = conv_encoder, encoder, assigner, decoder
        self.spec_aug_cfg = spec_aug_cfg
        self.ctc_fc = nn.Linear(encoder.d_output, decoder.d_output, bias=False)
        _xavier_all(self)

    _augment = Transformer._augment

    def forward(self, features, len_features, targets, threshold=0.95, noise=None):
        _assign_names(self)
        features, len_features = self._augment(features, len_features)
        def run():
            ctc3d, len_sequence, _num, num, logits = self._forward_impl(features, len_features, targets, threshold, noise)
            slots = _slots(self, "ctc", self, "num", self.decoder, "prj")
            return [ctc3d, _num, logits], slots, (len_sequence, num)
        (ctc3d, _num, logits), (len_sequence, num) = _taped(self, run)
        return ctc3d, len_sequence, _num, num, logits

    def _forward_impl(self, features, len_features, targets, threshold, noise):
        rec = _TAPE is not None
        conv, len_sequence = self.conv_encoder._impl(features, len_features)
        enc = self.encoder._impl(conv, len_sequence)
        ctc2d = _vocab_proj(self, "ctc", self.ctc_fc.weight, enc)
        asg_slot = {"g": None} if rec else None
        alpha_raw = self.assigner._impl(enc, len_sequence, asg_slot)
        if noise is None:
            noise = torch.rand(alpha_raw.size(0), device=alpha_raw.device)     # cif_model.py:47
        alpha, _num, num, scale = ops.cif_rescale_fwd(alpha_raw, targets, noise)   # cif_model.py:44-48
        if not rec:
            l = self.cif(enc.view3(), alpha, threshold=threshold)
            logits = self.decoder(l, targets)
            return ctc2d.view(enc.B, enc.L, -1), len_sequence, _num, num, logits
        # ---- recording: CIF with its backward closure (d_l arrives through cif_slot, d(_num) through the "num" slot) -----------------
        hidden = enc.view3()
        cur, rem, fire_idx, n_fire, n_label, tok = ops.cif_scan(alpha.float(), threshold, want_tok=True)
        umax = self.__dict__.get("_umax_hint")
        if umax is None:       # the reference's own host sync (`.max()` used as a tensor size, cif_model.py:98)
            stats = torch.stack([n_label.max(), n_fire.max()]).tolist()
            if stats[1] > stats[0]:
                raise RuntimeError("cif: a row fires %d times but max round(sum alpha) is %d (cif_model.py:100)" % (stats[1], stats[0]))
            umax = int(stats[0])
        # (trainer, longest target known: alpha was rescaled to sum to num +- 0.5 per row (cif_model.py:44-48), so round(sum alpha) = num
        # and a row fires at most num times - max_label_len IS the longest target, no read-back)
        self.last_fire = (fire_idx, n_fire, n_label)
        l = ops.cif_gather(hidden.float().contiguous(), cur, rem, fire_idx, n_fire, int(umax))
        cif_slot = {"g": None}
        num_slot = {"g": None, "shape": tuple(_num.shape)}
        self.__dict__.setdefault("_grad_slots", {})["num"] = num_slot

        def bw_cif():
            d_l = cif_slot["g"].reshape(l.shape)
            d_hidden, d_alpha = ops.cif_bwd(hidden, cur, rem, tok, n_fire, d_l)
            _acc(enc, d_hidden.view(enc.B * enc.L, -1))
            # backward of alpha = alpha_raw * (num_noise / sum(alpha_raw)) (cif_model.py:44-48), plus d(_num) from the quantity loss
            asg_slot["g"] = ops.cif_rescale_bwd(d_alpha, alpha_raw, scale, _num, num_slot["g"])
            cif_slot["g"] = None
            num_slot["g"] = None

        _TAPE.push(bw_cif, ())
        logits = self.decoder(l, targets, cif_slot)
        return ctc2d.view(enc.B, enc.L, -1), len_sequence, _num, num, logits

    def cif(self, hidden, alphas, threshold, log=False, max_label_len=None):
        """cif_model.py:57-106."""
        out, self.last_fire = cif_forward(hidden, alphas, threshold, max_label_len)
        return out

    @torch.no_grad()
    def recognize(self, input, input_length, char_list, args, threshold=0.95, target_num=None):
        """cif_model.py:108-131 - beam search for ONE utterance `input` [T, D] -> ([token lists incl. <sos>], [lengths])."""
        lens = torch.as_tensor(input_length, device=input.device).view(1)
        return self.batch_recognize(input.unsqueeze(0), lens, args.beam_size, args.nbest, threshold, target_num)[0]

    @torch.no_grad()
    def batch_recognize(self, features, len_features, beam_size, nbest=1, threshold=0.95, target_num=None):
        """`recognize` for a padded batch: conv front end, encoder, assigner, optional rescale of each row's weights to `target_num`
        (a number or one per utterance, cif_model.py:120-123), integrate-and-fire, then ONE batched beam search
        (Decoder_CIF.batch_recognize_beam).  Utterance b decodes round(sum alpha_b) steps - the length of the reference's zero-padded
        frame tensor for that utterance (cif_model.py:98-101).  -> [(token lists incl. <sos>, lengths)] per utterance, equal to
        `recognize` on each utterance alone when the batch is zero-padded (the loader's pad_list; the 'same' conv front end reads
        up to two frames past an utterance's end, conv_encoder.py:103-105)."""
        _assign_names(self)
        conv, len_sequence = self.conv_encoder._impl(features, len_features)
        enc = self.encoder._impl(conv, len_sequence)
        alpha = self.assigner._impl(enc, len_sequence, None)
        if target_num is not None and (torch.is_tensor(target_num) or target_num):
            num = torch.as_tensor(target_num, device=alpha.device, dtype=alpha.dtype).expand(alpha.shape[0])
            alpha = alpha * (num / alpha.sum(-1))[:, None]           # [B, L] rows
        l = self.cif(enc.view3(), alpha, threshold=threshold)
        n_label = self.last_fire[2]
        return self.decoder.batch_recognize_beam(l, n_label, beam_size, nbest)

    @classmethod
    def create_model(cls, args):
        """cif_model.py:133-163."""
        conv_encoder = Conv2dSubsample(d_input=args.d_input * args.LFR_m, d_model=args.d_model, n_layers=args.n_conv_layers)
        encoder = Encoder(d_input=args.d_model, n_layers=args.n_layers_enc, n_head=args.n_head, d_model=args.d_model,
                          d_inner=args.d_inner, dropout=args.dropout)
        assigner = Attention_Assigner(d_input=args.d_model, d_hidden=args.d_assigner_hidden, w_context=args.w_context,
                                      n_layers=args.n_assigner_layers)
        decoder = Decoder_CIF(sos_id=args.sos_id, n_tgt_vocab=args.vocab_size, n_layers=args.n_layers_dec, n_head=args.n_head,
                              d_model=args.d_model, d_inner=args.d_inner, dropout=args.dropout)
        return cls(conv_encoder, encoder, assigner, decoder, args.spec_aug_cfg)
