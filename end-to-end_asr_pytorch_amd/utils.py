"""Mask helpers with the reference's names (src/utils/utils.py:125-165).

The HIP path never materialises these tensors - kernels compare indices against lengths - but callers that build
masks for their own purposes get the same values the reference produces.  Pure index arithmetic, any device."""
import torch


def sequence_mask(lengths, maxlen=None, dtype=torch.float):
    if maxlen is None:
        maxlen = int(lengths.max())
    pos = torch.arange(1, maxlen + 1, device=lengths.device)[None, :]
    return (pos <= lengths[:, None]).type(dtype)


def get_subsequent_mask(seq):
    sz_b, len_s = seq.size()
    m = torch.triu(torch.ones((len_s, len_s), device=seq.device, dtype=torch.uint8), diagonal=1)
    return m.unsqueeze(0).expand(sz_b, -1, -1)


def get_attn_key_pad_mask(seq_k, seq_q, pad_idx):
    return seq_k.le(pad_idx).unsqueeze(1).expand(-1, seq_q.size(1), -1)


def get_attn_pad_mask(input_lengths, expand_length):
    pad_mask = sequence_mask(input_lengths) < 1.0
    return pad_mask.unsqueeze(1).expand(-1, expand_length, -1)


def pad_list(xs, pad_value, max_len=None):
    """src/utils/utils.py:5-14 (returns (padded, lengths) like the reference's current version)."""
    n_batch = len(xs)
    lengths = torch.tensor([x.size(0) for x in xs]).long()
    max_len = int(lengths.max()) if not max_len else max_len
    pad = xs[0].new(n_batch, max_len, *xs[0].size()[1:]).fill_(pad_value)
    for i in range(n_batch):
        pad[i, :xs[i].size(0)] = xs[i]
    return pad, lengths
