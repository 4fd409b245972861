"""Decoder.preprocess as one launch (csrc/norm_embed.hip: asr_decoder_targets; src/transformer/decoder.py:42-58) against the oracle's
restatement and the golden fixture: pad entries in the middle of a row, empty rows, rows longer than a wave, the overflow flag."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops
from oracle import asr_oracle as O

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def _check(tg, sos, eos):
    o_in, o_out = O.decoder_preprocess(tg, sos, eos)
    umax = o_in.shape[1] - 1
    ys_in, ys_out, in_len = ops.decoder_targets(torch.from_numpy(tg).to(DEV), sos, eos, umax)
    np.testing.assert_array_equal(ys_in.cpu().numpy(), o_in)
    np.testing.assert_array_equal(ys_out.cpu().numpy(), o_out)
    np.testing.assert_array_equal(in_len.cpu().numpy(), (o_in > 0).sum(1).astype(np.int32))


def test_decoder_targets_golden(golden_dir):
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    _check(z["targets"], 2, 3)
    dec = asr_amd.Decoder(2, 3, 50, 1, 2, 64, 128, dropout=0.0).to(DEV)
    ys_in, ys_out = dec.preprocess(torch.from_numpy(z["targets"]).to(DEV))      # (reads the longest length back itself)
    np.testing.assert_array_equal(ys_out.cpu().numpy(), z["targets_eos"])


@pytest.mark.parametrize("B,U,seed", [(1, 1, 0), (3, 4, 1), (32, 50, 2), (7, 64, 3), (5, 65, 4), (33, 200, 5), (130, 333, 6)])
def test_decoder_targets_random(B, U, seed):
    rng = np.random.default_rng(seed)
    tg = rng.integers(1, 4000, size=(B, U)).astype(np.int64)
    tg[rng.random((B, U)) < 0.3] = 0            # pad entries anywhere, not only at the tail (decoder.py:46 strips them all)
    if B > 2:
        tg[1] = 0                               # an empty target
        tg[2] = rng.integers(1, 4000, size=U)   # a full one
    _check(tg, 4232, 4233)


def test_decoder_targets_overflow_flag():
    tg = torch.tensor([[5, 0, 6, 7], [0, 0, 9, 0]], device=DEV)
    flag = torch.zeros(1, dtype=torch.int32, device=DEV)
    ys_in, ys_out, in_len = ops.decoder_targets(tg, 2, 3, 2, overflow=flag)     # W = 3: row 0 holds 3 tokens
    assert int(flag.item()) == 1
    assert ys_in.cpu().tolist() == [[2, 5, 6], [2, 9, 0]]
    assert ys_out.cpu().tolist() == [[5, 6, 3], [9, 3, 0]]
    flag.zero_()
    ops.decoder_targets(tg, 2, 3, 3, overflow=flag)
    assert int(flag.item()) == 0


@pytest.mark.parametrize("B,U,seed", [(1, 1, 0), (3, 4, 1), (32, 50, 2), (7, 64, 3), (5, 65, 4), (33, 200, 5)])
def test_decoder_cif_targets(B, U, seed):
    """asr_decoder_cif_targets (Decoder_CIF.preprocess, decoder.py:356-366) against the oracle's two lines (oracle/asr_oracle.py:
    decoder_cif_forward) and the module's torch expression: <sos> shifted in, positions whose target is pad zeroed, lengths."""
    rng = np.random.default_rng(seed)
    tg = rng.integers(1, 4000, size=(B, U)).astype(np.int64)
    lens = rng.integers(0, U + 1, size=B)
    for b in range(B):
        tg[b, lens[b]:] = 0                     # tail padding
    if B > 2:
        tg[2, U // 2] = 0                       # and a pad entry in the middle of a row (the reference masks by position)
    pad_mask = (tg > 0).astype(np.int64)
    ref = np.concatenate([np.full((B, 1), 4232, dtype=np.int64), tg[:, :-1]], 1) * pad_mask
    ys_in, in_len = ops.decoder_cif_targets(torch.from_numpy(tg).to(DEV), 4232)
    np.testing.assert_array_equal(ys_in.cpu().numpy(), ref)
    np.testing.assert_array_equal(in_len.cpu().numpy(), pad_mask.sum(1).astype(np.int32))
    dec = asr_amd.Decoder_CIF(4232, 4233, 1, 2, 64, 128, dropout=0.0)
    np.testing.assert_array_equal(dec.preprocess(torch.from_numpy(tg)).numpy(), ref)


def test_decoder_module_remembers_a_truncated_target():
    """Decoder._preprocess with a caller-supplied `umax` that is too small (a loader passing a wrong max_target_len inside a captured
    step, where nothing else would notice): the module's overflow word is set and target_overflow() reports it."""
    import asr_amd
    dec = asr_amd.Decoder(2, 3, 50, 1, 2, 64, 128).to(DEV)
    tg = torch.tensor([[5, 6, 7, 8, 0, 0], [9, 10, 0, 0, 0, 0]], device=DEV)
    dec._preprocess(tg, umax=4)
    assert not dec.target_overflow()
    dec._preprocess(tg, umax=3)
    assert dec.target_overflow()
