"""The input step in front of the path: low-frame-rate stacking, SpecAugment, frame-budget batching.

Reference: src/utils/data.py:28-110 (AudioDataset batching), :113-160 (LFRCollate / _collate_fn), :163-188 (load_inputs_and_targets),
:191-218 (build_LFR_features), src/utils/utils.py:168-194 (spec_aug).  Feature matrices are read from Kaldi ark files by `kaldi_ark`
(the reference goes through the third-party kaldi_io package).
"""
import ctypes

import numpy as np
import torch

from . import ops
from ._lib import check, lib


def build_LFR_features(inputs, m, n):
    """utils/data.py:191-218 for ONE utterance on the host: inputs [T, D] ndarray -> [ceil(T/n), m*D] (stack m frames, skip n; the
    last frame stands in for frames past the end)."""
    inputs = np.asarray(inputs)
    T = inputs.shape[0]
    idx = np.minimum(np.arange(int(np.ceil(T / n)))[:, None] * n + np.arange(m)[None, :], T - 1)
    return inputs[idx].reshape(idx.shape[0], -1)


def lfr_batch(padded_features, feature_lengths, m, n):
    """build_LFR_features of every utterance of a padded batch on the device: [B,T,D] f32, lengths [B] ->
    ([B, ceil(T/n), m*D], int32 lengths ceil(len/n)); rows past an utterance's stacked length are zero (pad_list's padding)."""
    ops._req_cuda(padded_features, feature_lengths)
    x = padded_features.float().contiguous()
    B, T, D = x.shape
    lens = ops.as_i32(feature_lengths, x.device)
    Tl = (T + n - 1) // n
    y = torch.empty((B, Tl, m * D), device=x.device, dtype=torch.float32)
    len_out = torch.empty(B, device=x.device, dtype=torch.int32)
    check(lib().asr_lfr_stack(ops._stream(), ops._p(x), ops._p(lens), B, T, D, int(m), int(n), ops._p(y), ops._p(len_out)), "asr_lfr_stack")
    return y, len_out


def spec_aug_draws(config, B, device, generator=None):
    """The uniform draws of utils.py:176-190 in the reference's call order: per mask `torch.rand(size=[B])` for the width, then for
    the start -> f32 [(n_freq_loops + n_time) * 2, B].  (Its frequency loop runs time_mask_num times, utils.py:177.)"""
    freq_mask_num, freq_mask_width, time_mask_num, time_mask_width = (int(i) for i in config.split("-"))
    n = 2 * time_mask_num
    return torch.stack([torch.rand(size=[B], device=device, generator=generator) for _ in range(2 * n)]) if n else torch.zeros((0, B), device=device)


def spec_aug(padded_features, feature_lengths, config, rand=None):
    """utils/utils.py:168-194, in place on `padded_features` [B,T,V] (f32, CUDA) like the reference; returns (padded_features,
    feature_lengths).  `rand`: the draws (spec_aug_draws) when the caller wants to fix them; drawn here otherwise."""
    ops._req_cuda(padded_features)
    freq_mask_num, freq_mask_width, time_mask_num, time_mask_width = (int(i) for i in config.split("-"))
    x = padded_features
    assert x.dtype == torch.float32 and x.is_contiguous(), "spec_aug works in place on a contiguous f32 batch"
    B, T, V = x.shape
    if rand is None:
        rand = spec_aug_draws(config, B, x.device)
    rand = rand.to(device=x.device, dtype=torch.float32).contiguous()
    assert rand.shape == (4 * time_mask_num, B)
    lens = ops.as_i32(feature_lengths, x.device)
    fmean = torch.empty((B, T), device=x.device, dtype=torch.float32)
    tsum = torch.empty((B, V), device=x.device, dtype=torch.float32)
    check(lib().asr_spec_aug(ops._stream(), ops._p(x), ops._p(lens), B, T, V, ops._p(rand), time_mask_num, freq_mask_width, time_mask_num,
                             time_mask_width, ops._p(fmean), ops._p(tsum)), "asr_spec_aug")
    return padded_features, feature_lengths


def _frames(item):
    return int(item[1]["input"][0]["shape"][0])


def _labels(item):
    return int(item[1]["output"][0]["shape"][0])


def _cut_by_count(ordered, batch_size, max_length_in, max_length_out):
    """fixed utterance count, divided by 1 + how many times the head utterance exceeds the length limits"""
    pos = 0
    while pos < len(ordered):
        head = ordered[pos]
        shrink = max(int(_frames(head) / max_length_in), int(_labels(head) / max_length_out))
        size = max(1, int(batch_size / (1 + shrink)))
        yield ordered[pos:pos + size]
        pos += size


def _cut_by_frames(ordered, budget):
    """utterances are added until the frame total reaches the budget (the one that crosses it is included)"""
    pos = 0
    while pos < len(ordered):
        total, stop = 0, pos
        while stop < len(ordered) and total < budget:
            total += _frames(ordered[stop])
            stop += 1
        yield ordered[pos:stop]
        pos = stop


def make_minibatches(utts, batch_size, max_length_in, max_length_out, num_batches=0, batch_frames=0):
    """The batching of AudioDataset (utils/data.py:48-110) from shape metadata alone.  `utts`: the 'utts' dict of an espnet-style
    data.json ({key: {'input': [{'shape': [T, D]}], 'output': [{'shape': [U, V]}]}}).  Utterances with T / U < 5 are dropped, the
    rest sorted long to short (stable: equal lengths keep their file order); minibatches are then cut by count or, with
    batch_frames > 0, by frame budget; num_batches > 0 keeps only the first few (the reference's debug switch).
    -> list of minibatches, each a list of (key, sample) - what AudioDataset.minibatch holds."""
    kept = [item for item in utts.items() if _frames(item) / _labels(item) >= 5.0]
    ordered = sorted(kept, key=_frames, reverse=True)
    cuts = _cut_by_frames(ordered, batch_frames) if batch_frames > 0 else _cut_by_count(ordered, batch_size, max_length_in, max_length_out)
    batches = list(cuts)
    return batches[:num_batches] if num_batches > 0 else batches


def shard_by_length(lengths, world):
    """Utterance indices per data-parallel rank with similar total frames: longest first, dealt in a snake
    (0..N-1, N-1..0, ...) so that no rank collects all the long ones; every rank gets the same count when len(lengths) % world == 0
    (the trainer's exact global-batch mean assumes equal counts).  -> list of `world` index lists."""
    order = sorted(range(len(lengths)), key=lambda i: -int(lengths[i]))
    shards = [[] for _ in range(world)]
    for pos, idx in enumerate(order):
        lap, slot = divmod(pos, world)
        shards[slot if lap % 2 == 0 else world - 1 - slot].append(idx)
    return shards


def load_inputs_and_targets(batch, token2idx, label_type="token", LFR_m=1, LFR_n=1, read_mat=None):
    """utils/data.py:163-188: one minibatch (list of (key, sample) from `make_minibatches`) -> (xs, ys): feature matrices read from
    sample['input'][0]['feat'] ("file.ark:offset"), LFR-stacked on the host when LFR_m / LFR_n != 1, utterances without labels
    dropped, the rest sorted long to short (stable), labels mapped through token2idx to int64 arrays."""
    from . import kaldi_ark
    read_mat = read_mat or kaldi_ark.read_mat
    xs = [read_mat(b[1]["input"][0]["feat"]) for b in batch]
    ys = [b[1]["output"][0][label_type].split() for b in batch]
    if LFR_m != 1 or LFR_n != 1:
        xs = [build_LFR_features(x, LFR_m, LFR_n) for x in xs]
    order = sorted((i for i in range(len(xs)) if len(ys[i]) > 0), key=lambda i: -len(xs[i]))
    if len(order) != len(xs):
        print("warning: Target sequences include empty token")
    return [xs[i] for i in order], [np.fromiter((token2idx[t] for t in ys[i]), dtype=np.int64) for i in order]


class LFRCollate(object):
    """utils/data.py:113-160 (LFRCollate + _collate_fn): the DataLoader's collate_fn.  `batch` is a list holding ONE minibatch
    (AudioDataset.__getitem__) -> (xs_pad f32 [N, Tmax, D] zero-padded, ilens int64 [N], ys_pad int64 [N, Umax] zero-padded)."""

    def __init__(self, token2idx, label_type, LFR_m=1, LFR_n=1, read_mat=None):
        self.token2idx, self.label_type, self.LFR_m, self.LFR_n, self.read_mat = token2idx, label_type, LFR_m, LFR_n, read_mat

    def __call__(self, batch):
        assert len(batch) == 1
        xs, ys = load_inputs_and_targets(batch[0], self.token2idx, self.label_type, self.LFR_m, self.LFR_n, self.read_mat)
        ilens = torch.from_numpy(np.array([x.shape[0] for x in xs], dtype=np.int64))
        xs_pad = torch.zeros((len(xs), max(x.shape[0] for x in xs), xs[0].shape[1]), dtype=torch.float32)
        for i, x in enumerate(xs):
            xs_pad[i, :x.shape[0]] = torch.from_numpy(np.asarray(x, dtype=np.float32))
        ys_pad = torch.zeros((len(ys), max(len(y) for y in ys)), dtype=torch.long)
        for i, y in enumerate(ys):
            ys_pad[i, :len(y)] = torch.from_numpy(y)
        return xs_pad, ilens, ys_pad
