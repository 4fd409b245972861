"""Mirrors of the masked-LM family (src/mask_lm/{Mask_LM,encoder,decoder,loss}.py: BERT-style pre-training over discrete units,
then CTC fine-tuning) on the same HIP kernels.

Same EncoderLayer stack as the speech models (mask_lm/encoder.py:4 imports transformer.encoder.EncoderLayer); new here are the
token input (embedding -> LayerNorm -> + PE), the token masking and the masked cross entropy with the reference's denominator.
"""
import torch
import torch.nn as nn

from . import ops
from .ctc_model import Decoder  # noqa: F401  (src/mask_lm/decoder.py is src/ctcModel/decoder.py: projection, then `* sequence_mask`)
from .loss import cal_loss
from .modules import (Act, EncoderLayer, PositionalEncoding, _Cached, _assign_names, _drop, _slots, _taped, _vocab_proj,
                      _xavier_all)
from . import modules


class Encoder(_Cached):
    """src/mask_lm/encoder.py:8-58 - token ids in: dropout(LayerNorm(Embedding(ids)) + PE), then the EncoderLayer stack."""

    def __init__(self, n_src, n_layers, n_head, d_model, d_inner, dropout=0.1):
        super().__init__()
        self.token_emb = nn.Embedding(n_src, d_model)
        self.n_layers, self.n_head, self.d_model, self.d_input, self.d_output, self.d_inner = n_layers, n_head, d_model, n_src, d_model, d_inner
        self.dropout_rate = dropout
        self.layer_norm_in = nn.LayerNorm(d_model)
        self.positional_encoding = PositionalEncoding(d_model)
        self.layer_stack = nn.ModuleList([EncoderLayer(d_model, d_inner, n_head, dropout=dropout) for _ in range(n_layers)])

    def _impl(self, ids, lens):
        B, L = ids.shape
        D = self.d_model
        rec = modules._TAPE is not None
        emb = self.token_emb
        zero_pe = torch.zeros((L, D), device=ids.device, dtype=torch.float32)
        e32, _ = ops.embed_pe(ids.contiguous(), emb.weight.detach().float(), zero_pe)          # the gather; LayerNorm comes before the PE
        dp = _drop(self, "dropout")   # mask_lm/encoder.py:48
        ln = self.layer_norm_in
        y32, y16, mean, rstd = ops.add_layernorm(e32, None, ln.weight, ln.bias, B, L, pe=self.positional_encoding.rows(L),
                                                 want_bf16=(modules.get_precision() == "bf16"), eps=ln.eps, save_stats=rec, drop_y=dp)
        x = Act(y32, y16, B, L)
        if rec:
            y0 = x

            def bw():
                ds, _ = ops.add_layernorm_bwd(y0.grad, e32, mean, rstd, ln.weight, None, B, L, ln.weight.grad, ln.bias.grad, drop_y=dp)
                y0.grad = None
                ops.embed_bwd(ids, ds, emb.weight.grad)

            modules._TAPE.push(bw, (emb.weight, ln.weight, ln.bias))
        masks = modules._prefetch_attn_masks([(layer.slf_attn, B, L, L) for layer in self.layer_stack], self.training, y32.device)
        for i, layer in enumerate(self.layer_stack):
            x = layer._impl(x, lens, attn_drop=masks[i])
        return x

    def forward(self, padded_input, input_lengths):
        return self._impl(padded_input, ops.as_i32(input_lengths, padded_input.device)).view3()


class Mask_LM(_Cached):
    """src/mask_lm/Mask_LM.py:5-63 - returns (logits_AE, logits, masked_index)."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        self.fc = nn.Linear(encoder.d_output, encoder.d_input, bias=False)
        _xavier_all(self)

    def token_mask(self, padded_input, p=0.05, M=10, rand=None):
        """Mask_LM.py:19-41.  rand: the reference's `torch.rand((B, T))` draws when the caller fixes them."""
        if rand is None:
            rand = torch.rand(padded_input.shape, device=padded_input.device)
        return ops.token_mask(padded_input, rand, p, M)

    def forward(self, padded_input, input_lengths, padded_target=None, mask_input=True, rand=None):
        _assign_names(self)
        if mask_input:
            masked_input, masked_index = self.token_mask(padded_input, rand=rand)
        else:
            masked_input, masked_index = padded_input, None
        want_dec = padded_target is not None
        B, L = masked_input.shape

        def run():
            lens = ops.as_i32(input_lengths, masked_input.device)
            enc = self.encoder._impl(masked_input, lens)
            outs = [_vocab_proj(self, "fc", self.fc.weight, enc).view(B, L, -1)]
            slots = _slots(self, "fc")
            if want_dec:
                # two projections read `enc`: the decoder's is recorded last, so its closure runs first and leaves its share in enc.grad
                outs.append(self.decoder._impl(enc, lens))
                slots = slots + _slots(self.decoder, "prj")
            return outs, slots, None
        outs, _ = _taped(self, run)
        return outs[0], (outs[1] if want_dec else None), masked_index

    @classmethod
    def create_model(cls, args):
        """Mask_LM.py:81-92."""
        encoder = Encoder(args.n_src, args.n_layers_enc, args.n_head, args.d_model, args.d_inner, dropout=args.dropout)
        return cls(encoder, Decoder(args.n_tgt, args.d_model))


class _CeMaskLossFn(torch.autograd.Function):
    @staticmethod
    def forward(ctx, logits2d, targets1d, counted1d, smoothing):
        loss2, row_loss, lse, tg = ops.ce_mask_loss_fwd(logits2d.detach(), targets1d, counted1d, smoothing)
        ctx.save = (logits2d.detach(), tg, float(smoothing), lse, loss2)
        return loss2[0].reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits2d, tg, smoothing, lse, loss2 = ctx.save
        return ops.ce_loss_bwd(logits2d, tg, smoothing, lse, loss2, gout), None, None, None


def cal_ce_mask_loss(logits, targets, mask, smoothing=0.0):
    """src/mask_lm/loss.py:5-32 - label-smoothed CE summed over every non-pad position, divided by the number of MASKED non-pad
    positions (the reference's denominator)."""
    V = logits.size(-1)
    logits2d = logits.reshape(-1, V)
    if logits2d.stride(1) != 1:
        logits2d = logits2d.contiguous()
    return _CeMaskLossFn.apply(logits2d, targets.contiguous().view(-1), mask.contiguous().view(-1), smoothing)


def cal_ctc_loss(logits, len_logits, gold, smoothing=0.0):
    """src/mask_lm/loss.py:35-45 (same arithmetic as src/ctcModel/loss.py)."""
    return cal_loss(logits, len_logits, gold, smoothing)
