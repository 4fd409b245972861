"""Greedy decoding on the device.

* `GreedyDecoder` / `ctc_greedy_decode`: src/ctcModel/ctc_infer.py:28-46,69-80 - per-frame argmax over the vocabulary, collapse
  repeats, drop blanks - as two row kernels (asr_argmax_rows, asr_ctc_greedy_reduce) instead of a Python loop over frames.
* `Decoder.batch_decode` / `Decoder.step` live in modules.py (they need the decoder's layers): src/transformer/decoder.py:98-164.
  The reference re-runs the whole decoder over the growing prefix at every step; here each step feeds ONE new token through the
  layers against per-layer K/V caches (self-attention) and the encoder-side K/V projected once for all steps and layers -
  identical values (the step's masks are causal-only, decoder.py:100-104, so position i never sees later tokens).
"""
import numpy as np
import torch

from . import ops


def ctc_greedy_decode(logits, lens, blank=None):
    """logits f32 [B, L, V] (last dim contiguous), lens int [B] -> (tokens int64 [B, L] zero-padded, n_tokens int32 [B]) on the device."""
    B, L, V = logits.shape
    x = logits if logits.dtype == torch.float32 else logits.float()
    if x.stride(2) != 1 or x.stride(0) != L * x.stride(1):
        x = x.contiguous()
    frames = ops.argmax_rows(torch.as_strided(x, (B * L, V), (x.stride(1), 1), x.storage_offset()))
    return ops.ctc_greedy_reduce(frames.view(B, L), lens, V - 1 if blank is None else blank)


class GreedyDecoder:
    """src/ctcModel/ctc_infer.py:69-80 (base class :10-66): `decoder(prob_tensor, frame_seq_len)` -> (int32 ndarray [B, max_len]
    zero-padded, list of lengths) - the return convention of `padding_list_seqs`."""

    def __init__(self, space_idx=1, blank_index=0):
        self.space_idx, self.blank_index = space_idx, blank_index

    def __call__(self, prob_tensor, frame_seq_len=None):
        return self.decode(prob_tensor, frame_seq_len)

    def decode(self, prob_tensor, frame_seq_len):
        B, L, _ = prob_tensor.shape
        if frame_seq_len is None:
            frame_seq_len = torch.full((B,), L, dtype=torch.int32, device=prob_tensor.device)
        tokens, n = ctc_greedy_decode(prob_tensor, frame_seq_len, self.blank_index)
        n_host = n.cpu().tolist()                      # one host read-back: the padded width is data dependent
        width = max(n_host) if n_host else 0
        return tokens[:, :width].cpu().numpy().astype(np.int32), n_host
