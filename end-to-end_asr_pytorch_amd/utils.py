"""Mask / padding helpers under the reference's names (src/utils/utils.py:5-14,125-165), for callers that build such tensors
themselves.  The HIP path never materialises masks - kernels compare indices against lengths - so nothing on the product path
imports this module; values match the reference's (tests/test_host_logic.py)."""
import torch


def sequence_mask(lengths, maxlen=None, dtype=torch.float):
    """[B, maxlen] with 1 where position < length."""
    n = int(lengths.max()) if maxlen is None else int(maxlen)
    steps = torch.arange(n, device=lengths.device)
    return (steps[None, :] < lengths[:, None]).to(dtype)


def get_subsequent_mask(seq):
    """[B, L, L] uint8, 1 strictly above the diagonal (a query may not see later keys)."""
    B, L = seq.shape
    idx = torch.arange(L, device=seq.device)
    return (idx[None, :] > idx[:, None]).to(torch.uint8)[None].expand(B, L, L)


def get_attn_key_pad_mask(seq_k, seq_q, pad_idx):
    """[B, Lq, Lk] bool, True on padded keys (token id <= pad_idx)."""
    return (seq_k <= pad_idx)[:, None, :].expand(seq_k.shape[0], seq_q.shape[1], seq_k.shape[1])


def get_attn_pad_mask(input_lengths, expand_length):
    """[B, expand_length, Lmax] bool, True on frames at or beyond each utterance's length."""
    keep = sequence_mask(input_lengths, dtype=torch.bool)
    return (~keep)[:, None, :].expand(keep.shape[0], expand_length, keep.shape[1])


def pad_list(xs, pad_value, max_len=None):
    """Stack variable-length tensors [len_i, ...] into [n, max_len, ...] filled with pad_value -> (padded, lengths)."""
    lengths = torch.as_tensor([int(x.shape[0]) for x in xs], dtype=torch.long)
    width = int(max_len) if max_len else int(lengths.max())
    out = torch.full((len(xs), width) + tuple(xs[0].shape[1:]), pad_value, dtype=xs[0].dtype, device=xs[0].device)
    for row, x in zip(out, xs):
        row[:x.shape[0]] = x
    return out, lengths
