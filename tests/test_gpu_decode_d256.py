"""The decode paths at the benchmark width (d_model = 256, h = 4) against the REFERENCE's outputs (tests/golden/g15_decode_d256.npz:
Decoder.batch_decode decoder.py:138-164, batch_beam_decode :166-234, CIF_Model.recognize cif_model.py:108-131).  At this width the bf16
product path runs its one-launch decode sub-layers (csrc/decode_blocks.hip), which the d_model = 64 fixtures cannot reach.
f32 mode: token-exact everywhere.  bf16: token-exact wherever the reference's own top-2 margin is above the rounding noise (the fixture
stores every greedy decision's margin; setting "a" has flat distributions and varied tokens, setting "b" sharp ones)."""
import argparse
import os

import numpy as np
import pytest
import torch

import asr_amd
from asr_amd import ops
from weights import crc_of, make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"
MARGIN = 0.3      # log-prob units; bf16 scores of these models are within ~0.05 of the fp32 ones


def _z(golden_dir):
    return np.load(os.path.join(golden_dir, "g15_decode_d256.npz"))


def _cfg(z):
    return {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}


def _model(z, tag, cls=None):
    sd = make_state_dict(names_shapes_from_json(z[tag + "_names_shapes"]), int(z[tag + "_seed"]))
    assert crc_of(sd) == int(z[tag + "_crc"])
    cls = cls or asr_amd.Conv_CTC_Transformer
    model = cls.create_model(argparse.Namespace(spec_aug_cfg=None, **_cfg(z)))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    with torch.no_grad():
        model.decoder.tgt_word_prj.weight.mul_(float(z[tag + "_scale"]))
    return model.to(DEV).eval()


def _agree_until_low_margin(pred, ref, margin):
    """rows of tokens equal to the reference's up to (excluding) the first decision whose top-2 margin is below MARGIN"""
    for b in range(ref.shape[0]):
        low = np.nonzero(margin[b] < MARGIN)[0]
        n = int(low[0]) if len(low) else ref.shape[1]
        np.testing.assert_array_equal(pred[b, :n], ref[b, :n], err_msg="row %d, first %d decisions" % (b, n))
    return sum(int((np.nonzero(margin[b] < MARGIN)[0][:1].tolist() or [ref.shape[1]])[0]) for b in range(ref.shape[0]))


@pytest.mark.parametrize("tag", ["a", "b"])
def test_greedy_decode_d256(golden_dir, tag, monkeypatch):
    z = _z(golden_dir)
    model = _model(z, tag)
    enc, enc_len = torch.from_numpy(z[tag + "_enc_out"]).to(DEV), torch.from_numpy(z[tag + "_enc_len"]).to(DEV)
    T = z[tag + "_greedy_preds"].shape[1]
    with asr_amd.precision("f32"):
        p, l, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=T)
        np.testing.assert_array_equal(p.cpu().numpy(), z[tag + "_greedy_preds"])
        np.testing.assert_array_equal(l.cpu().numpy(), z[tag + "_greedy_len"])
        model.decoder.eos_id = int(z[tag + "_eos2"])
        p, l, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=T)
        np.testing.assert_array_equal(p.cpu().numpy(), z[tag + "_greedy_preds_eos2"])
        np.testing.assert_array_equal(l.cpu().numpy(), z[tag + "_greedy_len_eos2"])
        model.decoder.eos_id = _cfg(z)["eos_id"]
        sc = model.decoder.step(torch.from_numpy(z[tag + "_step_prefix"]).to(DEV), enc, enc_len)
        np.testing.assert_allclose(sc.cpu().numpy(), z[tag + "_step_scores"], atol=3e-4, rtol=1e-4)
    # bf16: the one-launch sub-layers must be what runs, and their tokens are the reference's wherever the decision is not a coin toss
    calls = {"self": 0, "ffn": 0}
    o_self, o_ffn = ops.decode_self_attn, ops.decode_ffn
    monkeypatch.setattr(ops, "decode_self_attn", lambda *a, **k: (calls.__setitem__("self", calls["self"] + 1), o_self(*a, **k))[1])
    monkeypatch.setattr(ops, "decode_ffn", lambda *a, **k: (calls.__setitem__("ffn", calls["ffn"] + 1), o_ffn(*a, **k))[1])
    with asr_amd.precision("bf16"):
        p16, l16, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=T)
        s16 = model.decoder.step(torch.from_numpy(z[tag + "_step_prefix"]).to(DEV), enc, enc_len)
    assert calls["self"] > 0 and calls["ffn"] > 0, calls
    checked = _agree_until_low_margin(p16.cpu().numpy(), z[tag + "_greedy_preds"], z[tag + "_greedy_margin"])
    assert checked >= (8 if tag == "a" else 20), checked            # the comparison covers a real share of the decisions
    np.testing.assert_allclose(s16.cpu().numpy(), z[tag + "_step_scores"], atol=1.5e-1, rtol=3e-2)


@pytest.mark.parametrize("tag", ["a", "b"])
def test_beam_decode_d256(golden_dir, tag):
    z = _z(golden_dir)
    model = _model(z, tag)
    enc, enc_len = torch.from_numpy(z[tag + "_enc_out"]).to(DEV), torch.from_numpy(z[tag + "_enc_len"]).to(DEV)
    for case in str(z[tag + "_beam_cases"]).split("|"):
        beam, T, eos = (int(v) for v in case.split(","))
        k = "%s_beam_b%d_T%d_eos%d" % (tag, beam, T, eos)
        model.decoder.eos_id = eos
        with asr_amd.precision("f32"):
            p, l, sc = model.decoder.batch_beam_decode(enc, enc_len, beam_size=beam, max_decode_len=T)
        np.testing.assert_array_equal(p.cpu().numpy(), z[k + "_preds"])
        np.testing.assert_array_equal(l.cpu().numpy(), z[k + "_len"])
        np.testing.assert_allclose(sc.cpu().numpy(), z[k + "_scores"], rtol=1e-5, atol=3e-4)
        with asr_amd.precision("bf16"):
            p16, l16, s16 = model.decoder.batch_beam_decode(enc, enc_len, beam_size=beam, max_decode_len=T)
        ref_sc = z[k + "_scores"]
        np.testing.assert_allclose(s16.cpu().numpy()[:, 0], ref_sc[:, 0], atol=0.25, rtol=2e-2)          # the best beam's score
        for b in range(ref_sc.shape[0]):                                                             # its tokens, where it wins clearly
            if beam == 1 or ref_sc[b, 0] - ref_sc[b, 1] > 2 * MARGIN:
                np.testing.assert_array_equal(p16.cpu().numpy()[b, 0], z[k + "_preds"][b, 0])
    model.decoder.eos_id = _cfg(z)["eos_id"]


def test_cif_recognize_d256(golden_dir):
    z = _z(golden_dir)
    model = _model(z, "cif", asr_amd.CIF_Model)
    x, lens = torch.from_numpy(z["cif_x"]).to(DEV), torch.from_numpy(z["cif_lens"]).to(DEV)
    chars = ["c%d" % i for i in range(_cfg(z)["vocab_size"])]
    for case in str(z["cif_cases"]).split("|"):
        u, beam, nbest, tnum = (int(v) for v in case.split(","))
        k = "cif_u%d_b%d_n%d_t%d" % (u, beam, nbest, tnum)
        dargs = argparse.Namespace(beam_size=beam, nbest=nbest)
        T = int(lens[u])
        for prec in ("f32", "bf16"):
            with asr_amd.precision(prec):
                ys, ls = model.recognize(x[u, :T], lens[u:u + 1], chars, dargs, target_num=tnum)
            ref_y, ref_l = z[k + "_yseq"], z[k + "_len"]
            n_cmp = len(ref_l) if prec == "f32" else 1            # bf16: the best hypothesis (sharp setting: far ahead of the rest)
            for i in range(n_cmp):
                assert list(ys[i]) == [int(v) for v in ref_y[i][:ref_l[i]]], (k, prec, i)
                assert int(ls[i]) == int(ref_l[i])
