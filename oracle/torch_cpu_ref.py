"""Stock-PyTorch CPU baseline (TEST / MEASUREMENT INFRASTRUCTURE ONLY) of the joint CTC + attention training step.

SURVEY.md §8(d), last row: the CPU baseline timed beside the GPU number is *stock PyTorch CPU executing the same op
sequence as the reference*.  The reference's files cannot travel to the GPU box, so this module composes the same
`torch` / `torch.nn.functional` calls in the same order from a `state_dict` addressed by the reference's key names
(paths relative to /root/reference/):

    encoder      src/transformer/encoder.py:33-58,71-79    linear_in -> LayerNorm -> + PE -> dropout -> N x layer
    attention    src/transformer/attention.py:33-62,74-86  three Linear, view/permute/contiguous to (h*B, L, 64), mask.repeat,
                                                            bmm -> / sqrt(d_k) -> masked_fill(-inf) -> softmax -> dropout -> bmm,
                                                            un-permute, fc -> dropout -> + residual -> LayerNorm
    feed forward src/transformer/module.py:48-53            w_2(relu(w_1 x)) -> dropout -> + residual -> LayerNorm
    decoder      src/transformer/decoder.py:42-96,627-639  sos/eos bookkeeping, embedding + PE, N x (self, cross, ffn), projection
    conv         src/transformer/conv_encoder.py:101-126   F.pad, Conv2d stride (2,1) + ReLU, crop, flatten, affine
    losses       src/transformer/loss.py:5-48              F.log_softmax + F.ctc_loss(blank=V-1), label-smoothed CE
    step         src/transformer/solver.py:83-93            loss = ctc + ce ; backward   (torch autograd)

Only bench.py's `cpu_baseline` leg and tests/ import this file; the product path never does.  It is pinned like the numpy
oracle: tests/test_oracle_golden.py checks its logits, losses and gradients against the reference's own outputs (G0 / G1).
"""
import math

import torch
import torch.nn.functional as F


def _seq_mask(lengths, maxlen):
    return (torch.arange(1, maxlen + 1)[None, :] <= lengths[:, None]).float()           # utils.py:125-132


def _pad_mask(lengths, expand, maxlen):
    return (_seq_mask(lengths, maxlen) < 1.0).unsqueeze(1).expand(-1, expand, -1)       # utils.py:157-165


def _pe(sd, key, length, d_model):
    """module.py:7-32 - the `pe` buffer (taken from the state_dict when it carries one)."""
    if key in sd:
        return sd[key][:, :length]
    pos = torch.arange(0, length).unsqueeze(1).float()
    div = torch.exp(torch.arange(0, d_model, 2).float() * -(math.log(10000.0) / d_model))
    pe = torch.zeros(length, d_model)
    pe[:, 0::2] = torch.sin(pos * div)
    pe[:, 1::2] = torch.cos(pos * div)
    return pe.unsqueeze(0)


def _mha(sd, pfx, q, k, v, mask, h, p, train):
    B, Lq, _ = q.shape
    Lk = k.shape[1]
    res = q
    qh = F.linear(q, sd[pfx + "w_qs.weight"], sd[pfx + "w_qs.bias"]).view(B, Lq, h, 64)
    kh = F.linear(k, sd[pfx + "w_ks.weight"], sd[pfx + "w_ks.bias"]).view(B, Lk, h, 64)
    vh = F.linear(v, sd[pfx + "w_vs.weight"], sd[pfx + "w_vs.bias"]).view(B, Lk, h, 64)
    qh = qh.permute(2, 0, 1, 3).contiguous().view(-1, Lq, 64)
    kh = kh.permute(2, 0, 1, 3).contiguous().view(-1, Lk, 64)
    vh = vh.permute(2, 0, 1, 3).contiguous().view(-1, Lk, 64)
    m = mask.repeat(h, 1, 1)
    attn = torch.bmm(qh, kh.transpose(1, 2))
    attn = attn / math.sqrt(64.0)
    attn = attn.masked_fill(m, float("-inf"))
    attn = torch.softmax(attn, dim=2)
    attn = F.dropout(attn, p, train)
    out = torch.bmm(attn, vh)
    out = out.view(h, B, Lq, 64).permute(1, 2, 0, 3).contiguous().view(B, Lq, -1)
    out = F.dropout(F.linear(out, sd[pfx + "fc.weight"], sd[pfx + "fc.bias"]), p, train)
    D = res.shape[-1]
    return F.layer_norm(out + res, (D,), sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])


def _ffn(sd, pfx, x, p, train):
    out = F.linear(F.relu(F.linear(x, sd[pfx + "w_1.weight"], sd[pfx + "w_1.bias"])), sd[pfx + "w_2.weight"], sd[pfx + "w_2.bias"])
    out = F.dropout(out, p, train)
    return F.layer_norm(out + x, (x.shape[-1],), sd[pfx + "layer_norm.weight"], sd[pfx + "layer_norm.bias"])


def encoder(sd, x, lens, n_layers, h, p=0.0, train=False, pfx="encoder."):
    B, L, _ = x.shape
    npm = _seq_mask(lens, L).unsqueeze(-1)
    am = _pad_mask(lens, L, L)
    D = sd[pfx + "linear_in.weight"].shape[0]
    y = F.layer_norm(F.linear(x, sd[pfx + "linear_in.weight"], sd[pfx + "linear_in.bias"]), (D,),
                     sd[pfx + "layer_norm_in.weight"], sd[pfx + "layer_norm_in.bias"])
    y = F.dropout(y + _pe(sd, pfx + "positional_encoding.pe", L, D), p, train)
    for i in range(n_layers):
        lp = "%slayer_stack.%d." % (pfx, i)
        y = _mha(sd, lp + "slf_attn.", y, y, y, am, h, p, train) * npm
        y = _ffn(sd, lp + "pos_ffn.", y, p, train) * npm
    return y


def decoder(sd, targets, enc, enc_lens, n_layers, h, sos_id, eos_id, p=0.0, train=False, pfx="decoder."):
    ys = [t[t != 0] for t in targets]
    U = max(len(y) for y in ys) + 1
    ys_in = targets.new_zeros((len(ys), U))
    ys_out = targets.new_zeros((len(ys), U))
    for b, y in enumerate(ys):
        ys_in[b, 0], ys_in[b, 1:len(y) + 1] = sos_id, y
        ys_out[b, :len(y)], ys_out[b, len(y)] = y, eos_id
    npm = (ys_in > 0).unsqueeze(-1).float()
    sub = torch.triu(torch.ones((U, U), dtype=torch.uint8), diagonal=1).unsqueeze(0).expand(len(ys), -1, -1)
    keypad = ys_in.le(0).unsqueeze(1).expand(-1, U, -1)
    slf = (keypad.to(torch.uint8) + sub).gt(0)
    cross = _pad_mask(enc_lens, U, enc.shape[1])
    emb = sd[pfx + "tgt_word_emb.weight"]
    x = F.dropout(F.embedding(ys_in, emb) + _pe(sd, pfx + "positional_encoding.pe", U, emb.shape[1]), p, train)
    for i in range(n_layers):
        lp = "%slayer_stack.%d." % (pfx, i)
        x = _mha(sd, lp + "slf_attn.", x, x, x, slf, h, p, train) * npm
        x = _mha(sd, lp + "enc_attn.", x, enc, enc, cross, h, p, train) * npm
        x = _ffn(sd, lp + "pos_ffn.", x, p, train) * npm
    return F.linear(x, sd[pfx + "tgt_word_prj.weight"]), ys_out


def conv2d_subsample(sd, feats, lens, n_layers, pfx="conv_encoder."):
    B, T, D = feats.shape
    x = F.pad(feats, (0, 10, 0, 20)).unsqueeze(1)
    for i in range(n_layers):
        x = F.relu(F.conv2d(x, sd["%sconv.subsample/conv%d.weight" % (pfx, i)], sd["%sconv.subsample/conv%d.bias" % (pfx, i)],
                            stride=(2, 1)))
    dco = int(math.ceil(D / 2))
    x = x[:, :, :, :dco]
    Bc, C, Tc, Dc = x.shape
    x = x.permute(0, 2, 1, 3).contiguous().view(Bc, Tc, C * Dc)
    tl = T
    for _ in range(n_layers):
        lens = torch.ceil(lens.float() / 2).int()
        tl = int(math.ceil(tl / 2))
    x = x[:, :tl]
    return F.linear(x, sd[pfx + "affine.weight"], sd[pfx + "affine.bias"]), lens


def cal_ce_loss(logits, targets, smoothing):
    logits = logits.view(-1, logits.size(2))
    targets = targets.contiguous().view(-1)
    V = logits.size(1)
    one_hot = torch.zeros_like(logits).scatter(1, targets.long().view(-1, 1), 1)
    one_hot = one_hot * (1 - smoothing) + (1 - one_hot) * smoothing / V
    log_prb = F.log_softmax(logits, dim=1)
    npm = targets.ne(0)
    loss = -(one_hot * log_prb).sum(dim=1)
    return loss.masked_select(npm).sum() / npm.long().sum()


def cal_ctc_ce_loss(logits_ctc, len_ctc, logits_ce, targets, smoothing):
    V = logits_ctc.size(-1)
    tl = targets.ne(0).int().sum(1)
    lp = F.log_softmax(logits_ctc, dim=-1).transpose(0, 1)
    ctc = F.ctc_loss(lp, targets, len_ctc, tl, blank=V - 1)
    return ctc, cal_ce_loss(logits_ce, targets, smoothing)


def joint_step(sd, feats, lens, targets, cfg, conv_layers=0, p=0.0, train=False, smoothing=0.1, backward=True):
    """One pass of solver.py:83-93 (forward, loss = ctc + ce, backward).  sd: {reference key: leaf tensor requiring grad}.
    -> (ctc, ce, ctc_logits, logits); gradients land in the leaves' .grad."""
    h = cfg["n_head"]
    if conv_layers:
        x, l = conv2d_subsample(sd, feats, lens, conv_layers)
    else:
        x, l = feats, lens
    enc = encoder(sd, x, l, cfg["n_layers_enc"], h, p, train)
    ctc_logits = F.linear(enc, sd["ctc_fc.weight"])
    logits, teos = decoder(sd, targets, enc, l, cfg["n_layers_dec"], h, cfg["sos_id"], cfg["eos_id"], p, train)
    ctc, ce = cal_ctc_ce_loss(ctc_logits, l, logits, teos, smoothing)
    if backward:
        (ctc + ce).backward()
    return ctc, ce, ctc_logits, logits


def leaves(state_dict, requires_grad=True):
    """float32 CPU leaves keyed like the reference's state_dict (buffers `...pe` stay plain tensors)."""
    out = {}
    for k, v in state_dict.items():
        t = torch.as_tensor(v).detach().float().cpu().clone()
        if requires_grad and not k.endswith(".pe"):
            t.requires_grad_(True)
        out[k] = t
    return out


def ctc_op(B, T, U, V, iters, seed=0):
    """F.log_softmax + F.ctc_loss forward + backward at the stand-alone CTC shape of §8(d) -> mean ms per iteration."""
    import time
    g = torch.Generator().manual_seed(seed)
    logits = torch.randn(B, T, V, generator=g, requires_grad=True)
    targets = torch.randint(0, V - 1, (B, U), generator=g)
    il = torch.full((B,), T, dtype=torch.int32)
    tl = torch.full((B,), U, dtype=torch.int32)

    def once():
        logits.grad = None
        loss = F.ctc_loss(F.log_softmax(logits, dim=-1).transpose(0, 1), targets, il, tl, blank=V - 1)
        loss.backward()
    once()
    t0 = time.perf_counter()
    for _ in range(iters):
        once()
    return (time.perf_counter() - t0) / iters * 1e3
