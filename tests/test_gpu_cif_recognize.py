"""CIF inference on the device against the reference's own outputs (tests/golden/g13_cif_recognize.npz):
CIF_Model.recognize (src/transformer/cif_model.py:108-131), Decoder_CIF.recognize_beam / recognize_beam_cache / step_forward /
step_forward_cache (src/transformer/decoder.py:401-552)."""
import argparse
import os

import numpy as np
import pytest
import torch

import asr_amd
from weights import crc_of, make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "g13_cif_recognize.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    assert crc_of(sd) == int(z["crc"])
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return z, cfg, model.to(DEV).eval()


def cases(z):
    for case in str(z["cases"]).split("|"):
        u, beam, nbest, tnum = (int(v) for v in case.split(","))
        yield u, beam, nbest, tnum, "u%d_b%d_n%d_t%d" % (u, beam, nbest, tnum)


def test_recognize_is_token_exact_in_f32(golden_dir):
    """the whole inference path per utterance (conv, encoder, assigner, optional target_num rescale, CIF, beam search): the
    reference's n-best token lists and lengths, five utterance / beam / nbest / target_num settings"""
    z, cfg, model = load(golden_dir)
    chars = ["c%d" % i for i in range(cfg["vocab_size"])]
    with asr_amd.precision("f32"):
        for u, beam, nbest, tnum, tag in cases(z):
            T = int(z["lens"][u])
            x = torch.from_numpy(z["x"][u, :T]).to(DEV)
            ys, ls = model.recognize(x, torch.tensor([T], device=DEV), chars, argparse.Namespace(beam_size=beam, nbest=nbest),
                                     target_num=tnum or None)
            ref, ref_len = z["yseq_" + tag], z["len_" + tag]
            assert ls == ref_len.tolist(), tag
            for y, r, n in zip(ys, ref, ref_len):
                assert y == r[:n].tolist(), tag


def test_recognize_beam_on_the_reference_frames(golden_dir):
    """the beam search alone, fed the reference's integrated frames: f32 token-exact; bf16 returns hypotheses of the same shape
    whose best one scores (under the f32 model) within 1e-1 of the reference's best"""
    z, cfg, model = load(golden_dir)
    for u, beam, nbest, tnum, tag in cases(z):
        frames = torch.from_numpy(z["cif_" + tag]).to(DEV)
        args = argparse.Namespace(beam_size=beam, nbest=nbest)
        with asr_amd.precision("f32"):
            ys, ls = model.decoder.recognize_beam(frames, None, args)
            ys_c, ls_c = model.decoder.recognize_beam_cache(frames, None, args)
        ref, ref_len = z["yseq_" + tag], z["len_" + tag]
        assert ls == ref_len.tolist() and ys == [r[:n].tolist() for r, n in zip(ref, ref_len)], tag
        assert (ys_c, ls_c) == (ys, ls)
        with asr_amd.precision("bf16"):
            yb, lb = model.decoder.recognize_beam(frames, None, args)
        assert lb == ls and all(y[0] == cfg["sos_id"] for y in yb)

        def total(y):
            with asr_amd.precision("f32"):
                s = 0.0
                for t in range(len(y) - 1):
                    sc = model.decoder.step_forward(torch.tensor([y[:t + 1]], device=DEV), frames, t)
                    s += float(sc[0, y[t + 1]])
            return s
        assert total(yb[0]) >= total(ys[0]) - 1e-1, tag


def test_step_forward_and_step_forward_cache(golden_dir):
    """decoder.py:401-423 and :477-496 on a 3-token prefix for two hypotheses: scores and the [N, t + 1, n_layers, d] cache"""
    z, cfg, model = load(golden_dir)
    tag = list(cases(z))[-1][-1]
    frames = torch.from_numpy(np.repeat(z["cif_" + tag], 2, 0)).to(DEV)
    prefix = torch.from_numpy(z["step_prefix"]).to(DEV)
    for prec, tol in (("f32", dict(rtol=1e-4, atol=5e-5)), ("bf16", dict(rtol=5e-2, atol=6e-2))):
        with asr_amd.precision(prec):
            sc = model.decoder.step_forward(prefix, frames, 2)
            cache = torch.zeros((2, 0, cfg["n_layers_dec"], cfg["d_model"]), device=DEV)
            for t in range(3):
                sc_c, cache = model.decoder.step_forward_cache(prefix[:, :t + 1], frames, cache, t)
        np.testing.assert_allclose(sc.cpu().numpy(), z["step_scores"], **tol)
        np.testing.assert_allclose(sc_c.cpu().numpy(), z["step_scores_cache"], **tol)
        np.testing.assert_allclose(cache.cpu().numpy(), z["step_cache"], **tol)
        assert tuple(cache.shape) == (2, 3, cfg["n_layers_dec"], cfg["d_model"])


def test_recognize_beam_edge_shapes(golden_dir):
    """one integrated frame, beam of one, nbest larger than the beam, a beam the pruning kernel does not support"""
    z, cfg, model = load(golden_dir)
    tag = list(cases(z))[0][-1]
    frames = torch.from_numpy(z["cif_" + tag]).to(DEV)
    with asr_amd.precision("f32"):
        ys, ls = model.decoder.recognize_beam(frames[:, :1].contiguous(), None, argparse.Namespace(beam_size=3, nbest=7))
        assert ls == [2, 2, 2] and len({tuple(y) for y in ys}) == 3
        sc = model.decoder.step_forward(torch.tensor([[cfg["sos_id"]]], device=DEV), frames, 0)
        assert [y[1] for y in ys] == torch.topk(sc[0], 3).indices.tolist()
        g, lg = model.decoder.recognize_beam(frames, None, argparse.Namespace(beam_size=1, nbest=1))
        # a beam of one is greedy: every token is the argmax of step_forward on the prefix so far
        for t in range(len(g[0]) - 1):
            s = model.decoder.step_forward(torch.tensor([g[0][:t + 1]], device=DEV), frames, t)
            assert int(s[0].argmax()) == g[0][t + 1]
        with pytest.raises(Exception):
            model.decoder.recognize_beam(frames, None, argparse.Namespace(beam_size=9, nbest=1))


def test_batch_recognize_equals_per_utterance(golden_dir, monkeypatch):
    """the padded batch through ONE batched beam search returns, per utterance, what the per-utterance path returns (and so the
    reference's hypotheses): different frame counts per utterance, a per-utterance target_num, replayed and eager step"""
    z, cfg, model = load(golden_dir)
    x, lens = torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["lens"]).to(DEV)
    B = x.shape[0]
    # zero padding, as the loader's pad_list produces (the 'same' conv front end reads up to two frames past an utterance's end,
    # conv_encoder.py:103-105: the fixture's random padding would reach the last frames of the shorter rows)
    x = x * (torch.arange(x.shape[1], device=DEV)[None, :] < lens[:, None])[:, :, None]
    with asr_amd.precision("f32"):
        for beam, nbest, tnum in ((3, 2, None), (2, 2, 5), (4, 3, torch.tensor([9.0, 3.0, 6.0, 5.0][:B]))):
            res = {}
            for mode in ("1", "0"):
                monkeypatch.setenv("ASR_AMD_DECODE_GRAPH", mode)
                model.decoder.__dict__.pop("_beam_graph", None)
                res[mode] = model.batch_recognize(x, lens, beam, nbest, target_num=tnum)
            assert res["1"] == res["0"]
            for u in range(B):
                T = int(lens[u])
                t_u = None if tnum is None else (float(tnum[u]) if torch.is_tensor(tnum) else tnum)
                one = model.recognize(x[u, :T], lens[u:u + 1], None, argparse.Namespace(beam_size=beam, nbest=nbest), target_num=t_u)
                assert tuple(res["1"][u]) == tuple(one), (beam, u)
    # and against the fixture where a case coincides with a batch setting
    ref, ref_len = z["yseq_u0_b3_n2_t0"], z["len_u0_b3_n2_t0"]
    with asr_amd.precision("f32"):
        ys, ls = model.batch_recognize(x, lens, 3, 2)[0]
    assert ls == ref_len.tolist() and ys == [r[:n].tolist() for r, n in zip(ref, ref_len)]
