"""Mirrors of the pure-CTC model family (src/ctcModel/{ctc_model,encoder,decoder}.py) on the same HIP kernels."""
import torch.nn as nn

from . import ops
from .modules import (Act, Encoder as _TEncoder, EncoderLayer as _TEncoderLayer, MultiheadAttention, PositionalEncoding,
                      _Cached, _act, _assign_names, _slots, _taped, _vocab_proj, _xavier_all)


class MultiHeadAttention(MultiheadAttention):
    """src/ctcModel/attention.py:6-30 — same parameters, ctor argument order (n_head, d_model, d_k, d_v)."""

    def __init__(self, n_head, d_model, d_k, d_v, dropout=0.1):
        super().__init__(d_model, n_head, d_k, d_v, dropout)


class EncoderLayer(_TEncoderLayer):
    """src/ctcModel/encoder.py:71-92."""

    def __init__(self, d_model, d_inner, n_head, d_k, d_v, dropout=0.1):
        super().__init__(d_model, d_inner, n_head, dropout)
        if d_k != 64 or d_v != 64:
            raise NotImplementedError("d_k = d_v = 64 only")

    def forward(self, enc_input, non_pad_mask=None, slf_attn_mask=None, lengths=None):
        return super().forward(enc_input, non_pad_mask, slf_attn_mask, lengths), None


class Encoder(_TEncoder):
    """src/ctcModel/encoder.py:8-68 — explicit d_k/d_v/pe_maxlen; returns a 1-tuple."""

    def __init__(self, d_input, n_layers, n_head, d_k, d_v, d_model, d_inner, dropout=0.1, pe_maxlen=5000):
        super().__init__(d_input, n_layers, n_head, d_model, d_inner, dropout)
        if d_k != 64 or d_v != 64:
            raise NotImplementedError("d_k = d_v = 64 only")
        self.d_k, self.d_v, self.dim_output, self.pe_maxlen = d_k, d_v, d_model, pe_maxlen
        self.positional_encoding = PositionalEncoding(d_model, max_len=pe_maxlen)

    def forward(self, padded_input, input_lengths, return_attns=False):
        if return_attns:
            raise NotImplementedError("attention maps are never materialised by the fused kernel")
        return (super().forward(padded_input, input_lengths),)


class Decoder(_Cached):
    """src/ctcModel/decoder.py:7-40 — vocab projection then `logits *= sequence_mask`."""

    def __init__(self, n_tgt_vocab, d_input):
        super().__init__()
        self.n_tgt_vocab = self.dim_output = n_tgt_vocab
        self.tgt_word_prj = nn.Linear(d_input, n_tgt_vocab, bias=False)
        nn.init.xavier_normal_(self.tgt_word_prj.weight)

    def _impl(self, enc, lens, masking=True):
        # on the tape the projection's closure receives d(logits) through self._grad_slots["prj"]; rows t >= len carry no gradient
        # either way (the mask multiplies them by 0 and the CTC gradient is 0 there), so the mask needs no closure of its own
        logits = _vocab_proj(self, "prj", self.tgt_word_prj.weight, enc).view(enc.B, enc.L, -1)
        if masking:
            ops.mask_rows_(logits, lens)
        return logits

    def forward(self, encoder_padded_outputs, encoder_input_lengths, masking=True):
        lens = ops.as_i32(encoder_input_lengths, encoder_padded_outputs.device)
        return self._impl(_act(encoder_padded_outputs), lens, masking), encoder_input_lengths


class CTC_Model(nn.Module):
    """src/ctcModel/ctc_model.py:8-32."""

    def __init__(self, encoder, decoder):
        super().__init__()
        self.encoder, self.decoder = encoder, decoder
        _xavier_all(self)

    def forward(self, padded_input, input_lengths):
        _assign_names(self)

        def run():
            lens = ops.as_i32(input_lengths, padded_input.device)
            enc = self.encoder._impl(_act(padded_input), lens)
            logits = self.decoder._impl(enc, lens)
            return [logits], _slots(self.decoder, "prj"), None
        (logits,), _ = _taped(self, run)
        return logits, input_lengths

    def recognize(self, input, input_length, ctc_infer, args=None):
        """ctc_model.py:34-48 - one utterance [T, D] through the encoder and the projection, then `ctc_infer(logits, len_logits)`
        (e.g. asr_amd.GreedyDecoder)."""
        import torch
        with torch.no_grad():
            logits, len_logits = self.forward(input.unsqueeze(0), input_length)
        return ctc_infer(logits, len_logits)
