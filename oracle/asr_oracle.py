"""CPU oracle (TEST INFRASTRUCTURE ONLY) for the Speech-Transformer / CTC / CIF forward+loss path.

This file is a plain-numpy fp32 restatement of the arithmetic the reference
(eastonYi/end-to-end_asr_pytorch, mounted at /root/reference in the build container)
executes on its PyTorch-CPU path.  It exists only so that tests/, __graft_entry__.smoke()
and bench.py's `cpu_baseline` leg can check / time the HIP product path against it.
Nothing under end-to-end_asr_pytorch_amd/ may import it.

Pinning status: PINNED.  The reference ships no golden vectors of its own,
so the oracle is pinned against outputs of the reference itself, generated in the build
container by tests/golden/make_fixtures.py (which imports /root/reference/src) and committed
as tests/golden/*.npz; tests/test_oracle_golden.py replays every fixture through this file.
The CTC arithmetic lives in a third-party dependency (PyTorch aten `_ctc_loss`, the reference
pins only "PyTorch 1.5" in README.md:8); `ctc_loss` below restates the published
Graves-2006 alpha/beta recursion in log space and is pinned against torch 2.10 CPU
`F.ctc_loss` outputs captured in tests/golden/g2_ctc.npz.

Every function cites the reference file:line it follows (paths relative to /root/reference/).
All parameters are addressed by the reference's own state_dict key names.
"""
import contextlib
import math
import zlib

import numpy as np

F32 = np.float32
NEG_INF = F32(-np.inf)


# --------------------------------------------------------------------------------------
# dropout  (nn.Dropout in train mode: attention.py:59,83  module.py:51  encoder.py:48  decoder.py:83,385
#           attentionAssigner.py:35).  torch draws the Bernoulli mask from its global RNG stream, which no other
# implementation can reproduce; the product path instead derives the keep decision from a counter-based hash of
# the element index (include/asr_hip.h: asr_dropout_t).  This section restates that mask in numpy, so the
# oracle - and the REFERENCE itself, with nn.Dropout.forward patched to draw from here in
# tests/golden/make_fixtures.py (G6/G7) - can run "the reference's arithmetic under a given mask".
# --------------------------------------------------------------------------------------
_M64 = (1 << 64) - 1


def _splitmix64(x):
    x = (x + 0x9E3779B97F4A7C15) & _M64
    z = x
    z = ((z ^ (z >> 30)) * 0xBF58476D1CE4E5B9) & _M64
    z = ((z ^ (z >> 27)) * 0x94D049BB133111EB) & _M64
    return z ^ (z >> 31)


def dropout_site_keys(seed, name, call):
    """(key0, key1) for dropout site `name` (qualified nn.Dropout module name) on its call-th invocation (1-based)."""
    h = _splitmix64(seed & _M64)
    h = _splitmix64(h ^ zlib.crc32(name.encode()))
    h = _splitmix64(h ^ (call & _M64))
    return h & 0xFFFFFFFF, h >> 32


def _lowbias32(x):
    x = x.astype(np.uint32)
    x = x ^ (x >> np.uint32(16))
    x = x * np.uint32(0x7feb352d)
    x = x ^ (x >> np.uint32(15))
    x = x * np.uint32(0x846ca68b)
    return x ^ (x >> np.uint32(16))


def dropout_mask(shape3, thr16, key0, key1):
    """Multiplicative mask (0 or 65536/(65536-thr16)) for a tensor viewed as [N0, N1, N2] - asr_hip.h's definition."""
    N0, N1, N2 = (int(v) for v in shape3)
    n2h = (N2 + 1) // 2
    with np.errstate(over="ignore"):
        sub = _lowbias32(np.arange(N0, dtype=np.uint32) * np.uint32(0x9E3779B9) + np.uint32(key0))
        pair = np.arange(N1, dtype=np.uint32)[:, None] * np.uint32(n2h) + np.arange(n2h, dtype=np.uint32)[None, :]
        word = _lowbias32(pair[None] ^ sub[:, None, None]) ^ np.uint32(key1)
    r = np.stack([word & np.uint32(0xFFFF), word >> np.uint32(16)], -1).reshape(N0, N1, 2 * n2h)[:, :, :N2]
    scale = F32(65536.0) / F32(65536 - thr16)
    return (r >= np.uint32(thr16)).astype(F32) * scale


class DropoutCtx:
    """Mask source shared by the oracle functions below: keys from (seed, site name, per-site call count)."""

    def __init__(self, seed, p):
        self.seed, self.thr16, self.calls = int(seed), int(round(float(p) * 65536.0)), {}

    def mask(self, name, shape3):
        self.calls[name] = self.calls.get(name, 0) + 1
        k0, k1 = dropout_site_keys(self.seed, name, self.calls[name])
        return dropout_mask(shape3, self.thr16, k0, k1)


_DROPOUT = None


@contextlib.contextmanager
def dropout(seed, p):
    """with oracle.dropout(seed, p): ...  -> the forward functions below run in "train mode" under the hash masks."""
    global _DROPOUT
    old, _DROPOUT = _DROPOUT, (DropoutCtx(seed, p) if p > 0 else None)
    try:
        yield _DROPOUT
    finally:
        _DROPOUT = old


def _drop(x, name):
    """x [N0,N1,N2] -> dropout(x) at site `name` (identity outside a `dropout(...)` context)."""
    if _DROPOUT is None:
        return x
    return (x * _DROPOUT.mask(name, x.shape)).astype(F32)


# --------------------------------------------------------------------------------------
# masks  (src/utils/utils.py:125-165)
# --------------------------------------------------------------------------------------
def sequence_mask(lengths, maxlen=None):
    """src/utils/utils.py:125-132 — 1.0 where position < length, width = max(lengths)."""
    lengths = np.asarray(lengths).astype(np.int64)
    if maxlen is None:
        maxlen = int(lengths.max())
    pos = np.arange(1, maxlen + 1)[None, :]
    return (pos <= lengths[:, None]).astype(F32)


def get_attn_pad_mask(input_lengths, expand_length):
    """src/utils/utils.py:157-165 — True where the KEY position is padding; [B, Lq, Lk]."""
    pad = sequence_mask(input_lengths) < 1.0
    return np.broadcast_to(pad[:, None, :], (pad.shape[0], expand_length, pad.shape[1]))


def get_subsequent_mask(seq):
    """src/utils/utils.py:135-143 — strict upper triangle (future positions)."""
    b, n = seq.shape
    m = np.triu(np.ones((n, n), dtype=np.uint8), k=1)
    return np.broadcast_to(m[None], (b, n, n))


def get_attn_key_pad_mask(seq_k, seq_q, pad_idx=0):
    """src/utils/utils.py:146-154."""
    pm = seq_k <= pad_idx
    return np.broadcast_to(pm[:, None, :], (seq_k.shape[0], seq_q.shape[1], seq_k.shape[1]))


# --------------------------------------------------------------------------------------
# primitive layers
# --------------------------------------------------------------------------------------
def linear(x, w, b=None):
    """nn.Linear: y = x W^T + b, W is [out, in] row-major."""
    y = x.astype(F32) @ w.astype(F32).T
    if b is not None:
        y = y + b.astype(F32)
    return y.astype(F32)


def layer_norm(x, g, b, eps=1e-5):
    """nn.LayerNorm over the last dim, biased variance, eps=1e-5 (torch default)."""
    x = x.astype(F32)
    mu = x.mean(-1, keepdims=True, dtype=F32)
    xc = x - mu
    var = (xc * xc).mean(-1, keepdims=True, dtype=F32)
    return (xc / np.sqrt(var + F32(eps)) * g.astype(F32) + b.astype(F32)).astype(F32)


def positional_encoding(length, d_model):
    """src/transformer/module.py:7-32 — sin on even cols, cos on odd cols, no scaling."""
    pe = np.zeros((length, d_model), dtype=F32)
    position = np.arange(0, length, dtype=F32)[:, None]
    div_term = np.exp(np.arange(0, d_model, 2, dtype=F32) * F32(-(math.log(10000.0) / d_model))).astype(F32)
    pe[:, 0::2] = np.sin(position * div_term)
    pe[:, 1::2] = np.cos(position * div_term)
    return pe


def softmax_lastdim(x):
    m = x.max(-1, keepdims=True)
    e = np.exp(x - m)
    return (e / e.sum(-1, keepdims=True, dtype=F32)).astype(F32)


def log_softmax(x):
    x = x.astype(F32)
    m = x.max(-1, keepdims=True)
    s = x - m
    lse = np.log(np.exp(s).sum(-1, keepdims=True, dtype=F32))
    return (s - lse).astype(F32)


# --------------------------------------------------------------------------------------
# attention / FFN blocks
# --------------------------------------------------------------------------------------
def multihead_attention(sd, pfx, q, k, v, mask, n_head, d_k=64, d_v=64):
    """src/transformer/attention.py:33-62 (+ ScaledDotProductAttention :74-86), dropout=0.

    mask: bool [B, Lq, Lk], True = masked (-inf before softmax).  Post-LN residual block.
    """
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    residual = q
    qh = linear(q, sd[pfx + "w_qs.weight"], sd[pfx + "w_qs.bias"]).reshape(B, Lq, n_head, d_k).transpose(0, 2, 1, 3)
    kh = linear(k, sd[pfx + "w_ks.weight"], sd[pfx + "w_ks.bias"]).reshape(B, Lk, n_head, d_k).transpose(0, 2, 1, 3)
    vh = linear(v, sd[pfx + "w_vs.weight"], sd[pfx + "w_vs.bias"]).reshape(B, Lk, n_head, d_v).transpose(0, 2, 1, 3)
    attn = (qh @ kh.transpose(0, 1, 3, 2)) / F32(np.power(d_k, 0.5))
    if mask is not None:
        attn = np.where(mask[:, None, :, :], NEG_INF, attn)
    attn = softmax_lastdim(attn.astype(F32))
    if _DROPOUT is not None:   # attention.py:83 on the [h*B, Lq, Lk] tensor whose leading index is head*B + b (attention.py:43-49)
        a = attn.transpose(1, 0, 2, 3).reshape(n_head * B, Lq, Lk)
        attn = _drop(a, pfx + "attention.dropout").reshape(n_head, B, Lq, Lk).transpose(1, 0, 2, 3)
    out = (attn @ vh).transpose(0, 2, 1, 3).reshape(B, Lq, n_head * d_v)
    out = _drop(linear(out, sd[pfx + "fc.weight"], sd[pfx + "fc.bias"]), pfx + "dropout")   # attention.py:59
    return layer_norm(out + residual, sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])


def positionwise_ffn(sd, pfx, x):
    """src/transformer/module.py:48-53 — LN(W2 relu(W1 x + b1) + b2 + x)."""
    h = np.maximum(linear(x, sd[pfx + "w_1.weight"], sd[pfx + "w_1.bias"]), 0)
    o = _drop(linear(h, sd[pfx + "w_2.weight"], sd[pfx + "w_2.bias"]), pfx + "dropout")   # module.py:51
    return layer_norm(o + x, sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])


def encoder_layer(sd, pfx, x, non_pad_mask, slf_attn_mask, n_head):
    """src/transformer/encoder.py:71-79."""
    x = multihead_attention(sd, pfx + "slf_attn.", x, x, x, slf_attn_mask, n_head)
    x = x * non_pad_mask
    x = positionwise_ffn(sd, pfx + "pos_ffn.", x)
    return x * non_pad_mask


def encoder_forward(sd, pfx, padded_input, input_lengths, n_layers, n_head):
    """src/transformer/encoder.py:33-58 — LN(Linear(x)) + PE, then N layers."""
    non_pad_mask = sequence_mask(input_lengths)[:, :, None]
    L = padded_input.shape[1]
    slf_mask = get_attn_pad_mask(input_lengths, L)
    x = layer_norm(linear(padded_input, sd[pfx + "linear_in.weight"], sd[pfx + "linear_in.bias"]),
                   sd[pfx + "layer_norm_in.weight"], sd[pfx + "layer_norm_in.bias"])
    x = _drop((x + positional_encoding(L, x.shape[-1])[None]).astype(F32), pfx + "dropout")   # encoder.py:48
    for i in range(n_layers):
        x = encoder_layer(sd, f"{pfx}layer_stack.{i}.", x, non_pad_mask, slf_mask, n_head)
    return x.astype(F32)


# --------------------------------------------------------------------------------------
# conv front-ends
# --------------------------------------------------------------------------------------
def _conv2d_valid(x, w, b, stride):
    """x [B,C,T,D], w [O,C,kh,kw], valid padding, stride (st, sd)."""
    B, C, T, D = x.shape
    O, _, kh, kw = w.shape
    st, sd_ = stride
    To = (T - kh) // st + 1
    Do = (D - kw) // sd_ + 1
    out = np.zeros((B, O, To, Do), dtype=F32)
    for i in range(kh):
        for j in range(kw):
            patch = x[:, :, i:i + st * (To - 1) + 1:st, j:j + sd_ * (Do - 1) + 1:sd_]  # B,C,To,Do
            out += np.einsum("bctd,oc->botd", patch, w[:, :, i, j], optimize=True).astype(F32)
    return out + b[None, :, None, None]


def conv2d_subsample(sd, pfx, feats, feat_lengths, n_layers):
    """src/transformer/conv_encoder.py:101-126 (pad='same').

    right-pad freq +10 / time +20, n x (3x3 stride (2,1) valid conv + ReLU), crop freq to
    ceil(D/2), [B,C,T,D]->[B,T,C*D], crop time to ceil(T/2^n), affine.  Lengths: ceil(len/2) per layer.
    """
    B, T, D = feats.shape
    x = np.pad(feats.astype(F32), ((0, 0), (0, 20), (0, 10)))[:, None]
    for i in range(n_layers):
        x = np.maximum(_conv2d_valid(x, sd[f"{pfx}conv.subsample/conv{i}.weight"],
                                     sd[f"{pfx}conv.subsample/conv{i}.bias"], (2, 1)), 0)
    d_conv_out = int(math.ceil(D / 2))
    x = x[:, :, :, :d_conv_out]
    Bc, C, Tc, Dc = x.shape
    x = x.transpose(0, 2, 1, 3).reshape(Bc, Tc, C * Dc)
    lens = np.asarray(feat_lengths).astype(np.float64)
    tl = T
    for _ in range(n_layers):
        lens = np.ceil(lens / 2.0)
        tl = int(math.ceil(tl / 2.0))
    assert tl <= x.shape[1]
    x = x[:, :tl]
    out = linear(x, sd[pfx + "affine.weight"], sd[pfx + "affine.bias"])
    return out, lens.astype(np.int32)


def conv1d_stack(sd, pfx, feats, n_layers, w_context, name="assigner"):
    """src/transformer/conv_encoder.py:33-49 (pad='same'): right-pad time by n*w, valid k=w convs + ReLU, crop."""
    B, T, D = feats.shape
    x = np.pad(feats.astype(F32), ((0, 0), (0, n_layers * w_context), (0, 0)))
    for i in range(n_layers):
        w = sd[f"{pfx}conv.{name}/conv1d_{i}.weight"]  # [O, C, k]
        b = sd[f"{pfx}conv.{name}/conv1d_{i}.bias"]
        To = x.shape[1] - w.shape[2] + 1
        y = np.zeros((B, To, w.shape[0]), dtype=F32)
        for j in range(w.shape[2]):
            y += x[:, j:j + To, :] @ w[:, :, j].T
        x = np.maximum(y + b, 0).astype(F32)
    return x[:, :T]


def attention_assigner(sd, pfx, padded_input, input_lengths, n_layers, w_context):
    """src/transformer/attentionAssigner.py:25-40 — sigmoid(Linear(conv stack)) * length mask."""
    x = _drop(conv1d_stack(sd, pfx + "conv.", padded_input, n_layers, w_context), pfx + "dropout")   # attentionAssigner.py:35
    a = linear(x, sd[pfx + "linear.weight"], sd[pfx + "linear.bias"])[..., 0]
    a = (1.0 / (1.0 + np.exp(-a.astype(F32)))).astype(F32)
    return a * sequence_mask(input_lengths)


# --------------------------------------------------------------------------------------
# decoders
# --------------------------------------------------------------------------------------
def decoder_preprocess(targets, sos_id, eos_id):
    """src/transformer/decoder.py:42-58 — strip pad(0), prepend sos / append eos, re-pad with 0."""
    ys = [np.asarray(y)[np.asarray(y) != 0] for y in targets]
    n = max(len(y) for y in ys) + 1
    ys_in = np.zeros((len(ys), n), dtype=np.int64)
    ys_out = np.zeros((len(ys), n), dtype=np.int64)
    for i, y in enumerate(ys):
        ys_in[i, 0] = sos_id
        ys_in[i, 1:1 + len(y)] = y
        ys_out[i, :len(y)] = y
        ys_out[i, len(y)] = eos_id
    return ys_in, ys_out


def decoder_forward(sd, pfx, targets, enc_out, enc_lengths, n_layers, n_head, sos_id, eos_id):
    """src/transformer/decoder.py:60-96 and DecoderLayer :627-639."""
    ys_in, ys_out = decoder_preprocess(targets, sos_id, eos_id)
    non_pad = (ys_in > 0).astype(F32)[:, :, None]
    slf_mask = (get_attn_key_pad_mask(ys_in, ys_in).astype(np.uint8) + get_subsequent_mask(ys_in)) > 0
    U = ys_in.shape[1]
    cross_mask = get_attn_pad_mask(enc_lengths, U)
    d = sd[pfx + "tgt_word_emb.weight"].shape[1]
    x = _drop(sd[pfx + "tgt_word_emb.weight"][ys_in].astype(F32) + positional_encoding(U, d)[None], pfx + "dropout")   # decoder.py:83
    for i in range(n_layers):
        lp = f"{pfx}layer_stack.{i}."
        x = multihead_attention(sd, lp + "slf_attn.", x, x, x, slf_mask, n_head) * non_pad
        x = multihead_attention(sd, lp + "enc_attn.", x, enc_out, enc_out, cross_mask, n_head) * non_pad
        x = positionwise_ffn(sd, lp + "pos_ffn.", x) * non_pad
    return linear(x, sd[pfx + "tgt_word_prj.weight"]), ys_out


def decoder_cif_forward(sd, pfx, cif_out, target, n_layers, n_head, sos_id):
    """src/transformer/decoder.py:356-399 (Decoder_CIF.preprocess/forward)."""
    target = np.asarray(target).astype(np.int64)
    pad_mask = (target > 0).astype(np.int64)
    ys_in = np.concatenate([np.full((target.shape[0], 1), sos_id, dtype=np.int64), target[:, :-1]], 1) * pad_mask
    non_pad = (target > 0).astype(F32)[:, :, None]
    slf_mask = (get_attn_key_pad_mask(ys_in, ys_in).astype(np.uint8) + get_subsequent_mask(ys_in)) > 0
    U = ys_in.shape[1]
    d = sd[pfx + "tgt_word_emb.weight"].shape[1]
    emb = _drop(sd[pfx + "tgt_word_emb.weight"][ys_in].astype(F32) + positional_encoding(U, d)[None], pfx + "dropout")   # decoder.py:385
    x = linear(np.concatenate([cif_out, emb], -1), sd[pfx + "input_affine.weight"])
    for i in range(n_layers):
        x = encoder_layer(sd, f"{pfx}layer_stack.{i}.", x, non_pad, slf_mask, n_head)
    return linear(np.concatenate([cif_out, x], -1), sd[pfx + "tgt_word_prj.weight"])


# --------------------------------------------------------------------------------------
# CIF  (src/transformer/cif_model.py:57-106)
# --------------------------------------------------------------------------------------
def aten_row_sum_f32(x):
    """`x.sum(-1)` of a contiguous fp32 [.., n] tensor in the order torch's CPU kernel adds (aten/src/ATen/native/cpu/SumKernel.cpp,
    `cascade_sum` -> `vectorized_inner_sum` -> `row_sum` -> `multi_row_sum`; torch 1.5 - 2.10, 8-float vectors): four interleaved
    chains of 8-lane vector adds with a 4-level cascade (a level empties into the next every 2^level_power steps), the chains
    added 1, 2, 3 onto chain 0, then - starting from zero - the n % 8 tail elements and the 8 lanes, all sequentially; rows shorter
    than 8 take the same structure on scalars.  cif_model.py:95 ROUNDS this sum to the label count, so its last bit matters:
    a row whose sum sits within an ulp of k + 0.5 gets another count from any other order.  Pinned bit for bit on torch's own
    sums in tests/golden/g16_cif_label_count.npz (lengths 1 .. 2049; checked up to 20000 when the fixture was written)."""
    x = np.ascontiguousarray(np.asarray(x, dtype=F32))
    lead, n = x.shape[:-1], x.shape[-1]
    rows = x.reshape(-1, n)
    W = 8 if n >= 8 else 1
    vec = n // W
    V = rows[:, :vec * W].reshape(-1, vec, W)
    ilp, levels = 4, 4
    size_ilp = vec // ilp
    lp = max(4, (0 if size_ilp <= 1 else int(np.ceil(np.log2(size_ilp)))) // levels)
    step, mask = 1 << lp, (1 << lp) - 1
    acc = np.zeros((levels, rows.shape[0], ilp, W), F32)
    i = 0
    while i + step <= size_ilp:
        for _ in range(step):
            acc[0] = acc[0] + V[:, i * ilp:(i + 1) * ilp]
            i += 1
        for j in range(1, levels):
            acc[j] = acc[j] + acc[j - 1]
            acc[j - 1] = 0
            if i & (mask << (j * lp)):
                break
    while i < size_ilp:
        acc[0] = acc[0] + V[:, i * ilp:(i + 1) * ilp]
        i += 1
    for j in range(1, levels):
        acc[0] = acc[0] + acc[j]
    part = acc[0]                                    # [rows, ilp, W]
    for i2 in range(size_ilp * ilp, vec):
        part[:, 0] = part[:, 0] + V[:, i2]
    for k in range(1, ilp):
        part[:, 0] = part[:, 0] + part[:, k]
    lanes = part[:, 0]                               # [rows, W]
    if W == 1:
        return lanes[:, 0].reshape(lead)
    out = np.zeros(rows.shape[0], F32)
    for e in range(vec * W, n):
        out = out + rows[:, e]
    for v in range(W):
        out = out + lanes[:, v]
    return out.reshape(lead)


def cif(hidden, alphas, threshold=0.95, max_label_len=None):
    """Sequential integrate-and-fire in the reference's exact fp32 operation order.

    Returns (out [B,Umax,H], fire_idx list-of-arrays, n_label [B] = round(sum alpha)).
    Raises ValueError where the reference would (a row has more fires than max round(sum alpha)).
    """
    hidden = hidden.astype(F32)
    alphas = alphas.astype(F32)
    B, T, H = hidden.shape
    thr = F32(threshold)
    integrate = np.zeros(B, dtype=F32)
    frame = np.zeros((B, H), dtype=F32)
    fires = np.zeros((B, T), dtype=F32)
    frames = np.zeros((B, T, H), dtype=F32)
    one = F32(1.0)
    for t in range(T):
        alpha = alphas[:, t]
        dc = one - integrate
        integrate = integrate + alpha
        fires[:, t] = integrate
        fire = integrate > thr
        integrate = np.where(fire, integrate - one, integrate).astype(F32)
        cur = np.where(fire, dc, alpha).astype(F32)
        rem = (alpha - cur).astype(F32)
        frame = frame + cur[:, None] * hidden[:, t, :]
        frames[:, t] = frame
        frame = np.where(fire[:, None], rem[:, None] * hidden[:, t, :], frame).astype(F32)
    # torch.round is round-half-to-even, same as np.round; the sum in torch's own fp32 order (aten_row_sum_f32, fixture G16)
    n_label = np.round(aten_row_sum_f32(alphas)).astype(np.int32)
    umax = int(n_label.max()) if max_label_len is None else int(max_label_len)
    out = np.zeros((B, umax, H), dtype=F32)
    fire_idx = []
    for b in range(B):
        idx = np.nonzero(fires[b] > thr)[0]
        fire_idx.append(idx.astype(np.int32))
        if len(idx) > umax:
            raise ValueError(f"cif: row {b} fires {len(idx)} > max_label_len {umax}")
        out[b, :len(idx)] = frames[b, idx]
    return out, fire_idx, n_label


# --------------------------------------------------------------------------------------
# losses
# --------------------------------------------------------------------------------------
def _logaddexp3(a, b, c):
    m = np.maximum(np.maximum(a, b), c)
    msafe = np.where(np.isneginf(m), F32(0), m)
    with np.errstate(divide="ignore"):
        r = np.log(np.exp(a - msafe) + np.exp(b - msafe) + np.exp(c - msafe)) + msafe
    return np.where(np.isneginf(m), NEG_INF, r).astype(F32)


def ctc_loss(log_probs, targets, input_lengths, target_lengths, blank, want_grad=False):
    """Graves alpha/beta recursion in log space — what `F.ctc_loss(lp[T,B,V], targets[B,Umax], in_len,
    tgt_len, blank)` computes at src/transformer/loss.py:42-43 and src/ctcModel/loss.py:10-11.

    log_probs: [B, T, V] (batch-major; the reference transposes to [T,B,V] for torch).
    Returns nll [B] (inf for infeasible alignments: zero_infinity=False) and, if want_grad, the
    gradient of sum_b nll_b wrt log_probs... expressed as torch does, i.e. d nll_b / d logits
    under the assumption log_probs = log_softmax(logits):  exp(lp) - exp(log(sum_s a_t(s) b_t(s)) + nll - lp).
    """
    lp = log_probs.astype(F32)
    B, T, V = lp.shape
    nll = np.zeros(B, dtype=F32)
    grad = np.zeros_like(lp) if want_grad else None
    for b in range(B):
        Tb = int(input_lengths[b])
        U = int(target_lengths[b])
        S = 2 * U + 1
        ext = np.full(S, blank, dtype=np.int64)
        ext[1::2] = np.asarray(targets[b][:U]).astype(np.int64)
        # skip transition s-2 -> s allowed where ext[s] != blank and ext[s] != ext[s-2]
        can_skip = np.zeros(S, dtype=bool)
        can_skip[2:] = (ext[2:] != blank) & (ext[2:] != ext[:-2])
        la = np.full((Tb, S), NEG_INF, dtype=F32)
        if Tb > 0:
            la[0, 0] = lp[b, 0, blank]
            if S > 1:
                la[0, 1] = lp[b, 0, ext[1]]
        for t in range(1, Tb):
            prev = la[t - 1]
            p1 = np.concatenate([[NEG_INF], prev[:-1]]).astype(F32)
            p2 = np.concatenate([[NEG_INF, NEG_INF], prev[:-2]]).astype(F32)
            p2 = np.where(can_skip, p2, NEG_INF)
            la[t] = _logaddexp3(prev, p1, p2) + lp[b, t, ext]
        if Tb == 0:
            ll = F32(0.0) if S == 1 else NEG_INF  # torch: empty input, loss 0 only for empty target
        elif S > 1:
            ll = np.logaddexp(la[Tb - 1, S - 1], la[Tb - 1, S - 2]).astype(F32)
        else:
            ll = la[Tb - 1, S - 1]
        nll[b] = -ll
        if want_grad and Tb > 0:
            lb = np.full((Tb, S), NEG_INF, dtype=F32)
            lb[Tb - 1, S - 1] = lp[b, Tb - 1, blank]
            if S > 1:
                lb[Tb - 1, S - 2] = lp[b, Tb - 1, ext[S - 2]]
            can_skip_fwd = np.zeros(S, dtype=bool)  # s -> s+2 allowed
            can_skip_fwd[:-2] = can_skip[2:]
            for t in range(Tb - 2, -1, -1):
                nxt = lb[t + 1]
                n1 = np.concatenate([nxt[1:], [NEG_INF]]).astype(F32)
                n2 = np.concatenate([nxt[2:], [NEG_INF, NEG_INF]]).astype(F32)
                n2 = np.where(can_skip_fwd, n2, NEG_INF)
                lb[t] = _logaddexp3(nxt, n1, n2) + lp[b, t, ext]
            lab = la + lb  # [Tb, S]
            g = np.full((Tb, V), NEG_INF, dtype=F32)
            for s in range(S):
                g[:, ext[s]] = np.logaddexp(g[:, ext[s]], lab[:, s])
            with np.errstate(over="ignore", invalid="ignore"):
                grad[b, :Tb] = np.exp(lp[b, :Tb]) - np.exp(g + nll[b] - lp[b, :Tb])
    return (nll, grad) if want_grad else nll


def ctc_loss_mean(logits, targets, input_lengths, blank=None, want_grad=False):
    """F.log_softmax + F.ctc_loss(reduction='mean', zero_infinity=False) as called at
    src/transformer/loss.py:41-43: mean_b( nll_b / max(tgt_len_b, 1) ); blank = V-1;
    target_lengths = count of non-zero ids per row (loss.py:40)."""
    logits = logits.astype(F32)
    V = logits.shape[-1]
    blank = V - 1 if blank is None else blank
    targets = np.asarray(targets)
    tgt_len = (targets != 0).sum(1).astype(np.int64)
    lp = log_softmax(logits)
    r = ctc_loss(lp, targets, input_lengths, tgt_len, blank, want_grad=want_grad)
    nll = r[0] if want_grad else r
    denom = np.maximum(tgt_len, 1).astype(F32)
    loss = (nll / denom).mean(dtype=F32)
    if not want_grad:
        return loss, nll
    scale = (F32(1.0) / (denom * F32(len(nll))))[:, None, None]
    g = r[1] * scale
    # log_softmax backward: g - softmax * sum_v g  (row sums are ~0 analytically)
    g = g - np.exp(lp) * g.sum(-1, keepdims=True)
    return loss, nll, g.astype(F32)


def cal_ce_loss(logits, targets, smoothing=0.0):
    """src/transformer/loss.py:5-31 — un-normalised label smoothing, mean over targets != 0."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V).astype(F32)
    tg = np.asarray(targets).reshape(-1).astype(np.int64)
    lp = log_softmax(lg)
    non_pad = tg != 0
    n_word = int(non_pad.sum())
    if smoothing > 0.0:
        eps = F32(smoothing)
        one_hot = np.zeros_like(lg)
        one_hot[np.arange(len(tg)), tg] = 1
        one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / F32(V)
        loss = -(one_hot * lp).sum(1, dtype=F32)
    else:
        loss = -lp[np.arange(len(tg)), tg]
    return F32(loss[non_pad].sum(dtype=F32) / F32(n_word))


def cal_ctc_ce_loss(logits_ctc, len_logits_ctc, logits_ce, targets, smoothing=0.0):
    """src/transformer/loss.py:34-48."""
    ctc, _ = ctc_loss_mean(logits_ctc, targets, len_logits_ctc)
    return ctc, cal_ce_loss(logits_ce, targets, smoothing)


def cal_ctc_qua_ce_loss(logits_ctc, len_logits_ctc, _number, number, logits_ce, targets, smoothing=0.0):
    """src/transformer/loss.py:51-61."""
    qua = F32(((np.asarray(_number, dtype=F32) - np.asarray(number, dtype=F32)) ** 2).mean(dtype=F32))
    ctc, ce = cal_ctc_ce_loss(logits_ctc, len_logits_ctc, logits_ce, targets, smoothing)
    return qua, ctc, ce


# --------------------------------------------------------------------------------------
# model wrappers  (return-tuple conventions of SURVEY §8a row 18)
# --------------------------------------------------------------------------------------
def transformer_forward(sd, features, len_features, targets, cfg):
    """src/transformer/transformer.py:21-35 - the attention-only family: (logits, targets_eos) (+ the encoder output)."""
    enc = encoder_forward(sd, "encoder.", features, len_features, cfg["n_layers_enc"], cfg["n_head"])
    logits, teos = decoder_forward(sd, "decoder.", targets, enc, len_features, cfg["n_layers_dec"],
                                   cfg["n_head"], cfg["sos_id"], cfg["eos_id"])
    return logits, teos, enc


def ctc_transformer_forward(sd, features, len_features, targets, cfg):
    """src/transformer/transformer.py:108-124 — (ctc_len, ctc_logits, (logits, targets_eos))."""
    enc = encoder_forward(sd, "encoder.", features, len_features, cfg["n_layers_enc"], cfg["n_head"])
    ctc_logits = linear(enc, sd["ctc_fc.weight"])
    logits, teos = decoder_forward(sd, "decoder.", targets, enc, len_features, cfg["n_layers_dec"],
                                   cfg["n_head"], cfg["sos_id"], cfg["eos_id"])
    return np.asarray(len_features), ctc_logits, (logits, teos), enc


def conv_ctc_transformer_forward(sd, features, len_features, targets, cfg):
    """src/transformer/transformer.py:135-153 — (ctc_logits, len, logits, targets_eos) (+ intermediates)."""
    conv_out, lens = conv2d_subsample(sd, "conv_encoder.", features, len_features, cfg["n_conv_layers"])
    enc = encoder_forward(sd, "encoder.", conv_out, lens, cfg["n_layers_enc"], cfg["n_head"])
    ctc_logits = linear(enc, sd["ctc_fc.weight"])
    logits, teos = decoder_forward(sd, "decoder.", targets, enc, lens, cfg["n_layers_dec"],
                                   cfg["n_head"], cfg["sos_id"], cfg["eos_id"])
    return ctc_logits, lens, logits, teos, conv_out, enc


def cif_model_forward(sd, features, len_features, targets, cfg, noise, threshold=0.95):
    """src/transformer/cif_model.py:24-55 with the U(0,1) noise tensor supplied (cif_model.py:47)."""
    conv_out, lens = conv2d_subsample(sd, "conv_encoder.", features, len_features, cfg["n_conv_layers"])
    enc = encoder_forward(sd, "encoder.", conv_out, lens, cfg["n_layers_enc"], cfg["n_head"])
    ctc_logits = linear(enc, sd["ctc_fc.weight"])
    alpha = attention_assigner(sd, "assigner.", enc, lens, cfg["n_assigner_layers"], cfg["w_context"])
    _num = alpha.sum(-1, dtype=F32)
    num = (np.asarray(targets) > 0).astype(F32).sum(-1, dtype=F32)
    num_noise = num + np.asarray(noise, dtype=F32) - F32(0.5)
    alpha = alpha * (num_noise / _num)[:, None]
    l, fire_idx, _ = cif(enc, alpha, threshold)
    logits = decoder_cif_forward(sd, "decoder.", l, targets, cfg["n_layers_dec"], cfg["n_head"], cfg["sos_id"])
    return ctc_logits, lens, _num, num, logits, alpha, l, fire_idx


def ctc_model_forward(sd, padded_input, input_lengths, cfg):
    """src/ctcModel/ctc_model.py:21-32: encoder, vocab projection, logits *= sequence_mask
    (src/ctcModel/decoder.py:29-36)."""
    enc = encoder_forward(sd, "encoder.", padded_input, input_lengths, cfg["n_layers_enc"], cfg["n_head"])
    logits = linear(enc, sd["decoder.tgt_word_prj.weight"])
    logits = logits * sequence_mask(input_lengths)[:, :, None]
    return logits, np.asarray(input_lengths)


# --------------------------------------------------------------------------------------
# decoding helpers, input step, mask_lm  
# --------------------------------------------------------------------------------------
def ctc_greedy(frame_tokens, lengths, blank):
    """src/ctcModel/ctc_infer.py:37-46 - keep a frame label that is not blank and differs from the previous frame's label."""
    out = []
    for row, n in zip(np.asarray(frame_tokens), np.asarray(lengths)):
        keep, prev = [], None
        for tok in row[:int(n)]:
            if tok != blank and tok != prev:
                keep.append(int(tok))
            prev = tok
        out.append(keep)
    return out


def decoder_step(sd, pfx, prefixs, enc_out, enc_lengths, n_layers, n_head):
    """src/transformer/decoder.py:98-120 - log-softmax scores of the next token; every prefix position counts (no pad mask)."""
    prefixs = np.asarray(prefixs).astype(np.int64)
    U = prefixs.shape[1]
    slf_mask = get_subsequent_mask(prefixs) > 0
    cross_mask = get_attn_pad_mask(enc_lengths, U)
    d = sd[pfx + "tgt_word_emb.weight"].shape[1]
    x = sd[pfx + "tgt_word_emb.weight"][prefixs].astype(F32) + positional_encoding(U, d)[None]
    for i in range(n_layers):
        lp = f"{pfx}layer_stack.{i}."
        x = multihead_attention(sd, lp + "slf_attn.", x, x, x, slf_mask, n_head)
        x = multihead_attention(sd, lp + "enc_attn.", x, enc_out, enc_out, cross_mask, n_head)
        x = positionwise_ffn(sd, lp + "pos_ffn.", x)
    return log_softmax(linear(x[:, -1], sd[pfx + "tgt_word_prj.weight"]))


def _topk(x, k):
    """torch.topk(sorted=True) over the last dim; ties to the lower index -> (values, indices)"""
    idx = np.argsort(-x, axis=-1, kind="stable")[..., :k]
    return np.take_along_axis(x, idx, -1), idx


def batch_beam_decode(sd, pfx, enc_out, enc_lengths, n_layers, n_head, sos_id, eos_id, beam_size, max_decode_len):
    """src/transformer/decoder.py:166-234.  Restated with its quirks: the initial scores are [0, -1e10, ...] (`inf = 1e10`, :10),
    every step recomputes the whole prefix, `finished` and `len_decoded` belong to the beam SLOT (they are not re-gathered with the
    beams at :204-209), finished beams keep being extended.  -> (preds [B, beam, steps], len_decoded [B, beam], scores [B, beam])"""
    B = len(enc_lengths)
    enc = np.repeat(np.asarray(enc_out)[:, None], beam_size, 1).reshape(B * beam_size, *np.asarray(enc_out).shape[1:])
    lens = np.repeat(np.asarray(enc_lengths)[:, None], beam_size, 1).reshape(-1)
    preds = np.full((B * beam_size, 1), sos_id, np.int64)
    len_decoded = np.ones(B * beam_size, np.int64)
    scores = np.tile(np.array([0.0] + [-1e10] * (beam_size - 1), F32), B)
    finished = np.zeros(B * beam_size, bool)
    base = np.repeat(np.arange(B), beam_size)
    for _ in range(max_decode_len):
        z = log_softmax(decoder_step(sd, pfx, preds, enc, lens, n_layers, n_head))
        next_scores, next_preds = _topk(z, beam_size)
        cand = (scores[:, None] + next_scores).astype(F32).reshape(B, beam_size * beam_size)
        _, k_idx = _topk(cand, beam_size)
        k_idx = base * beam_size * beam_size + k_idx.reshape(-1)
        scores = cand.reshape(-1)[k_idx]
        nxt = next_preds.reshape(-1)[k_idx]
        preds = np.concatenate([preds[k_idx // beam_size], nxt[:, None]], 1)
        finished = finished | (nxt == eos_id)
        len_decoded = len_decoded + 1 - finished.astype(np.int64)
        if finished.all():
            break
    len_decoded = len_decoded - (1 - finished.astype(np.int64))
    preds = preds[:, 1:]
    s_sorted, order = _topk(scores.reshape(B, beam_size), beam_size)
    order = base * beam_size + order.reshape(-1)
    return preds[order].reshape(B, beam_size, -1), len_decoded[order].reshape(B, beam_size), s_sorted


def decoder_cif_step_forward(sd, pfx, ys, cif_out, t, n_layers, n_head):
    """src/transformer/decoder.py:401-423 (Decoder_CIF.step_forward): log-softmax scores [N, V] of the token after the prefix
    `ys` int64 [N, t + 1], the decoder fed the first t + 1 integrated frames; causal mask only, no pad mask."""
    ys = np.asarray(ys).astype(np.int64)
    U = ys.shape[1]
    slf_mask = get_subsequent_mask(ys) > 0
    ones = np.ones((ys.shape[0], U, 1), F32)
    d = sd[pfx + "tgt_word_emb.weight"].shape[1]
    emb = sd[pfx + "tgt_word_emb.weight"][ys].astype(F32) + positional_encoding(U, d)[None]
    frames = np.asarray(cif_out, F32)[:, :t + 1]
    x = linear(np.concatenate([frames, emb], -1), sd[pfx + "input_affine.weight"])
    for i in range(n_layers):
        x = encoder_layer(sd, f"{pfx}layer_stack.{i}.", x, ones, slf_mask, n_head)
    x = np.concatenate([frames, x], -1)
    return log_softmax(linear(x[:, -1], sd[pfx + "tgt_word_prj.weight"]))


def decoder_cif_step_forward_cache(sd, pfx, ys, cif_out, dec_cache, t, n_layers, n_head):
    """src/transformer/decoder.py:477-496 with EncoderLayer.forward_cache (encoder.py:81-87): only the last position goes through
    each layer (its query against the layer's whole input, no mask), the earlier positions' layer outputs come from `dec_cache`
    [N, t, n_layers, d] -> (scores [N, V], new cache [N, t + 1, n_layers, d])"""
    ys = np.asarray(ys).astype(np.int64)
    U = ys.shape[1]
    d = sd[pfx + "tgt_word_emb.weight"].shape[1]
    emb = sd[pfx + "tgt_word_emb.weight"][ys].astype(F32) + positional_encoding(U, d)[None]
    frames = np.asarray(cif_out, F32)[:, :t + 1]
    x = linear(np.concatenate([frames, emb], -1), sd[pfx + "input_affine.weight"])
    new_cache = []
    for i in range(n_layers):
        lp = f"{pfx}layer_stack.{i}."
        last = multihead_attention(sd, lp + "slf_attn.", x[:, -1:], x, x, None, n_head)
        last = positionwise_ffn(sd, lp + "pos_ffn.", last)
        x = np.concatenate([np.asarray(dec_cache, F32)[:, :, i], last], 1)
        new_cache.append(x[:, :, None])
    new_cache = np.concatenate(new_cache, 2)
    x = np.concatenate([frames, x], -1)
    return log_softmax(linear(x[:, -1], sd[pfx + "tgt_word_prj.weight"])), new_cache


def decoder_cif_recognize_beam(sd, pfx, cif_out, n_layers, n_head, sos_id, beam_size, nbest):
    """src/transformer/decoder.py:425-475 (Decoder_CIF.recognize_beam), one utterance: exactly `maxlen` = number of integrated
    frames steps (no <eos> handling), every live hypothesis extended by its `beam` best tokens, the candidates (hypothesis-major,
    rank-minor) stable-sorted by accumulated score and cut to `beam`.  -> (token lists incl. <sos>, their lengths, scores)"""
    cif_out = np.asarray(cif_out, F32)
    hyps = [(F32(0.0), [sos_id])]
    for i in range(cif_out.shape[1]):
        kept = []
        for score, yseq in hyps:
            z = decoder_cif_step_forward(sd, pfx, np.array([yseq]), cif_out, i, n_layers, n_head)
            best, ids = _topk(z, beam_size)
            for j in range(beam_size):
                kept.append((F32(score + best[0, j]), yseq + [int(ids[0, j])]))
        order = sorted(range(len(kept)), key=lambda c: -kept[c][0])      # stable, like Python's sorted(reverse=True) on the scores
        hyps = [kept[c] for c in order[:beam_size]]
    order = sorted(range(len(hyps)), key=lambda c: -hyps[c][0])[:min(len(hyps), nbest)]
    return [hyps[c][1] for c in order], [len(hyps[c][1]) for c in order], [hyps[c][0] for c in order]


def cif_model_recognize(sd, feats, cfg, beam_size, nbest, threshold=0.95, target_num=None):
    """src/transformer/cif_model.py:108-131 (CIF_Model.recognize), one utterance feats [T, D]"""
    feats = np.asarray(feats, F32)[None]
    conv_out, lens = conv2d_subsample(sd, "conv_encoder.", feats, np.array([feats.shape[1]]), cfg["n_conv_layers"])
    enc = encoder_forward(sd, "encoder.", conv_out, lens, cfg["n_layers_enc"], cfg["n_head"])
    alpha = attention_assigner(sd, "assigner.", enc, lens, cfg["n_assigner_layers"], cfg["w_context"])
    if target_num:
        alpha = alpha * (F32(target_num) / alpha.sum(-1, dtype=F32))[:, None]
    l, _, _ = cif(enc, alpha, threshold)
    return decoder_cif_recognize_beam(sd, "decoder.", l, cfg["n_layers_dec"], cfg["n_head"], cfg["sos_id"], beam_size, nbest), l, alpha


def lfr(inputs, m, n):
    """src/utils/data.py:191-218 - stack m frames every n frames; the last frame repeats past the end."""
    T = len(inputs)
    rows = []
    for i in range(int(np.ceil(T / n))):
        rows.append(np.concatenate([inputs[min(i * n + j, T - 1)] for j in range(m)]))
    return np.stack(rows)


def spec_aug(x, lengths, config, rand):
    """src/utils/utils.py:168-194 with the uniform draws handed in (rand [(loops * 2) * 2, B] in the reference's call order).
    Both loops run time_mask_num times (utils.py:177, :186); means come from the unmasked features."""
    _, fw, tn, tw = (int(v) for v in config.split("-"))
    x = np.array(x, dtype=F32)
    B, T, V = x.shape
    fmean = x.mean(-1, dtype=F32)
    tmean = (x.sum(1, dtype=F32) / np.asarray(lengths, dtype=F32)[:, None]).astype(F32)
    r = np.asarray(rand, dtype=F32)
    for k in range(tn):
        fs = (F32(fw) * r[2 * k]).astype(np.int64)
        f0 = ((V - fs).astype(F32) * r[2 * k + 1]).astype(np.int64)
        for b in range(B):
            x[b, :, f0[b]:f0[b] + fs[b]] = fmean[b][:, None]
    for k in range(tn):
        ts = (F32(tw) * r[2 * (tn + k)]).astype(np.int64)
        t0 = ((np.asarray(lengths) - ts).astype(F32) * r[2 * (tn + k) + 1]).astype(np.int64)
        for b in range(B):
            x[b, t0[b]:t0[b] + ts[b], :] = tmean[b][None, :]
    return x


def token_mask(ids, rand, p=0.05, M=10):
    """src/mask_lm/Mask_LM.py:19-41 - the keep mask ANDed with M circular left shifts of itself; masked ids become 0.
    -> (masked ids, masked positions)"""
    keep = np.asarray(rand, dtype=F32) > F32(p)
    shifted = keep.copy()
    for _ in range(M):
        shifted = np.concatenate([shifted[:, 1:], shifted[:, :1]], 1)
        keep = keep & shifted
    return np.where(keep, ids, 0), ~keep


def cal_ce_mask_loss(logits, targets, mask, smoothing=0.0):
    """src/mask_lm/loss.py:5-32 - the smoothed CE of every non-pad row summed, divided by the count of masked non-pad rows."""
    V = logits.shape[-1]
    lg = logits.reshape(-1, V).astype(F32)
    tg = np.asarray(targets).reshape(-1).astype(np.int64)
    lp = log_softmax(lg)
    eps = F32(smoothing)
    one_hot = np.zeros_like(lg)
    one_hot[np.arange(len(tg)), tg] = 1
    one_hot = one_hot * (1 - eps) + (1 - one_hot) * eps / F32(V)
    loss = -(one_hot * lp).sum(1, dtype=F32)
    non_pad = tg != 0
    n_word = int((non_pad & np.asarray(mask).reshape(-1)).sum())
    return F32(loss[non_pad].sum(dtype=F32) / F32(n_word))


def mask_lm_forward(sd, ids, lengths, n_layers, n_head, with_decoder=False):
    """src/mask_lm/Mask_LM.py:43-63 after the masking: encoder (mask_lm/encoder.py:33-58: LN(Embedding) + PE, EncoderLayer stack),
    fc, optionally the masked projection of mask_lm/decoder.py:19-38.  -> (logits_AE, logits or None)"""
    pfx = "encoder."
    non_pad_mask = sequence_mask(lengths)[:, :, None]
    L = ids.shape[1]
    slf_mask = get_attn_pad_mask(lengths, L)
    x = layer_norm(sd[pfx + "token_emb.weight"][ids], sd[pfx + "layer_norm_in.weight"], sd[pfx + "layer_norm_in.bias"])
    x = _drop((x + positional_encoding(L, x.shape[-1])[None]).astype(F32), pfx + "dropout")
    for i in range(n_layers):
        x = encoder_layer(sd, f"{pfx}layer_stack.{i}.", x, non_pad_mask, slf_mask, n_head)
    logits_AE = linear(x, sd["fc.weight"])
    logits = None
    if with_decoder:
        logits = linear(x, sd["decoder.tgt_word_prj.weight"]) * sequence_mask(lengths)[:, :, None]
    return logits_AE, logits


# --------------------------------------------------------------------------------------
# harness semantics
# --------------------------------------------------------------------------------------
def noam_lr(step, k, d_model, warmup):
    """src/transformer/optimizer.py:24-29."""
    return k * d_model ** (-0.5) * min(step ** (-0.5), step * warmup ** (-1.5))


def adam_step(p, g, m, v, step, lr, betas=(0.9, 0.98), eps=1e-9):
    """torch.optim.Adam (no weight decay, no amsgrad) as configured at src/transformer/train.py:166-170."""
    b1, b2 = betas
    m = b1 * m + (1 - b1) * g
    v = b2 * v + (1 - b2) * g * g
    mhat = m / (1 - b1 ** step)
    vhat = v / (1 - b2 ** step)
    return (p - lr * mhat / (np.sqrt(vhat) + eps)).astype(F32), m.astype(F32), v.astype(F32)
