"""Greedy decoding on the device against the reference's own outputs (tests/golden/g9_decode.npz):
Decoder.batch_decode / step (src/transformer/decoder.py:98-164), Conv_CTC_Transformer.batch_recognize (transformer.py:172-185),
ctcModel's GreedyDecoder (ctc_infer.py:28-46,69-80)."""
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


def load(golden_dir):
    z = np.load(os.path.join(golden_dir, "g9_decode.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    assert crc_of(sd) == int(z["crc"])
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    model = asr_amd.Conv_CTC_Transformer.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return z, cfg, model.to(DEV).eval()


def test_batch_decode_is_token_exact_in_f32(golden_dir):
    z, cfg, model = load(golden_dir)
    enc, enc_len = torch.from_numpy(z["enc_out"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV)
    with asr_amd.precision("f32"):
        preds, n, extra = model.decoder.batch_decode(enc, enc_len, max_decode_len=12)
        np.testing.assert_array_equal(preds.cpu().numpy(), z["preds"])
        np.testing.assert_array_equal(n.cpu().numpy(), z["len_decoded"])
        assert extra.numel() == 0
        # rows that finish at different steps (the reference's `finished` / `len_decoded` bookkeeping, decoder.py:154-161)
        for eos in (14, 39):
            model.decoder.eos_id = eos
            p, l, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=12)
            np.testing.assert_array_equal(p.cpu().numpy(), z["preds_eos%d" % eos])
            np.testing.assert_array_equal(l.cpu().numpy(), z["len_eos%d" % eos])
        model.decoder.eos_id = cfg["eos_id"]
        # end to end from the features; the third argument is max_decode_len (transformer.py:183)
        p, l, _ = model.batch_recognize(torch.from_numpy(z["x"]).to(DEV), torch.from_numpy(z["lens"]).to(DEV), 5)
        np.testing.assert_array_equal(p.cpu().numpy(), z["preds_rec5"])
        np.testing.assert_array_equal(l.cpu().numpy(), z["len_rec5"])


def test_step_scores_and_cached_decode_agree_with_the_full_recompute(golden_dir):
    z, cfg, model = load(golden_dir)
    enc, enc_len = torch.from_numpy(z["enc_out"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV)
    with asr_amd.precision("f32"):
        scores = model.decoder.step(torch.from_numpy(z["step_prefix"]).to(DEV), enc, enc_len)
    np.testing.assert_allclose(scores.cpu().numpy(), z["step_scores"], atol=2e-4, rtol=1e-4)
    # bf16 product path: the same tokens here too (the fixture's argmax margins are far above bf16 noise) and scores within tolerance
    with asr_amd.precision("bf16"):
        s16 = model.decoder.step(torch.from_numpy(z["step_prefix"]).to(DEV), enc, enc_len)
        p16, n16, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=12)
    np.testing.assert_allclose(s16.cpu().numpy(), z["step_scores"], atol=8e-2, rtol=2e-2)
    assert (p16.cpu().numpy() == z["preds"]).mean() >= 0.9


def test_ctc_greedy_decoder(golden_dir):
    z, cfg, model = load(golden_dir)
    dec = asr_amd.GreedyDecoder(space_idx=-1, blank_index=6)
    tok, n = dec.decode(torch.from_numpy(z["syn_logits"]).to(DEV), torch.from_numpy(z["syn_len"]).to(DEV))
    assert n == z["syn_tokens_len"].tolist()
    np.testing.assert_array_equal(tok, z["syn_tokens"])
    assert tok.dtype == np.int32
    # the model's own CTC head, blank = V - 1
    dec = asr_amd.GreedyDecoder(space_idx=-1, blank_index=cfg["vocab_size"] - 1)
    tok, n = dec(torch.from_numpy(z["ctc_logits"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV))
    assert n == z["ctc_lens"].tolist()
    np.testing.assert_array_equal(tok, z["ctc_tokens"])


def test_argmax_and_log_softmax_rows():
    g = torch.Generator().manual_seed(0)
    x = torch.randn(37, 4234, generator=g)
    x[3, 100] = x[3, 4000] = 9.0          # a tie: the lower index wins (torch.argmax on CPU)
    x[5, :] = -1.5                        # all equal
    buf = torch.zeros(37, 4240)
    buf[:, :4234] = x
    got = ops.argmax_rows(buf.to(DEV)[:, :4234])
    np.testing.assert_array_equal(got.cpu().numpy(), torch.argmax(x, -1).numpy())
    assert int(got[3]) == 100 and int(got[5]) == 0
    ls = ops.log_softmax_rows(buf.to(DEV)[:, :4234])
    np.testing.assert_allclose(ls.cpu().numpy(), torch.log_softmax(x, -1).numpy(), atol=2e-5, rtol=1e-5)


def test_graph_replayed_decode_step_equals_the_eager_step(golden_dir, monkeypatch):
    """Decoder.batch_decode replays its per-token step from a hipGraph (position, cache slot, key length and finished flags live in
    device memory); with ASR_AMD_DECODE_GRAPH=0 the same step functions are queued eagerly.  Same tokens, lengths and stop step."""
    z, cfg, model = load(golden_dir)
    enc, enc_len = torch.from_numpy(z["enc_out"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV)
    out = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ASR_AMD_DECODE_GRAPH", mode)
        model.decoder.__dict__.pop("_decode_graph", None)
        res = []
        for eos in (cfg["eos_id"], 14, 39):
            model.decoder.eos_id = eos
            for T in (12, 3, 20):
                p, l, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=T)
                res.append((p.cpu().numpy(), l.cpu().numpy()))
        assert (model.decoder.__dict__["_decode_graph"]["graphs"] is not None) == (mode == "1")
        out[mode] = res
    model.decoder.eos_id = cfg["eos_id"]
    for (p1, l1), (p0, l0) in zip(out["1"], out["0"]):
        np.testing.assert_array_equal(p1, p0)
        np.testing.assert_array_equal(l1, l0)
    # the cached graph is reused for a second batch of the same shape (new encoder outputs are copied into its input buffer)
    monkeypatch.setenv("ASR_AMD_DECODE_GRAPH", "1")
    p_a, _, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=20)
    dg = model.decoder.__dict__["_decode_graph"]
    assert dg["graphs"] is not None
    a0 = p_a.cpu().numpy().copy()
    p_b, _, _ = model.decoder.batch_decode(enc.flip(0).contiguous(), enc_len.flip(0).contiguous(), max_decode_len=20)
    assert model.decoder.__dict__["_decode_graph"] is dg
    np.testing.assert_array_equal(p_a.cpu().numpy(), a0)        # (the first result is not the replay's scratch)
    np.testing.assert_array_equal(p_b.cpu().numpy(), a0[::-1])


def test_batch_beam_decode_matches_the_reference(golden_dir):
    """decoder.py:166-234 against the reference's own beams (G12: five beam-size / eos settings, beams that finish at different
    steps, the slot-bound `finished` / `len_decoded` quirk): tokens and lengths exact, scores to fp32 tolerance."""
    z = np.load(os.path.join(golden_dir, "g12_beam_decode.npz"))
    _, cfg, model = load(golden_dir)                       # same seeded weights as G9
    enc, enc_len = torch.from_numpy(z["enc_out"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV)
    with asr_amd.precision("f32"):
        for case in str(z["cases"]).split("|"):
            beam, T, eos = (int(v) for v in case.split(","))
            tag = "b%d_T%d_eos%d" % (beam, T, eos)
            model.decoder.eos_id = eos
            p, l, sc = model.decoder.batch_beam_decode(enc, enc_len, beam_size=beam, max_decode_len=T)
            np.testing.assert_array_equal(p.cpu().numpy(), z["preds_" + tag])
            np.testing.assert_array_equal(l.cpu().numpy(), z["len_" + tag])
            np.testing.assert_allclose(sc.cpu().numpy(), z["scores_" + tag], rtol=1e-5, atol=2e-4)
    model.decoder.eos_id = cfg["eos_id"]
    # top-k kernel against torch.topk on rows without ties, and its tie order (index order)
    g = torch.Generator().manual_seed(5)
    x = torch.randn(37, 4234, generator=g).to(DEV)
    v, i = asr_amd.ops.topk_rows(x, 5)
    tv, ti = torch.topk(x, 5, sorted=True)
    np.testing.assert_array_equal(i.cpu().numpy(), ti.cpu().numpy())
    np.testing.assert_array_equal(v.cpu().numpy(), tv.cpu().numpy())
    t = torch.tensor([[1.0, 3.0, 3.0, 2.0, 3.0, -1.0]], device=DEV)
    v, i = asr_amd.ops.topk_rows(t, 4)
    assert i.cpu().tolist() == [[1, 2, 4, 3]] and v.cpu().tolist() == [[3.0, 3.0, 3.0, 2.0]]


def test_batch_decode_smallest_shapes(golden_dir, monkeypatch):
    """One decode step, one utterance, zero steps - replayed and eager forms agree; the first token equals the full-length run's."""
    z, cfg, model = load(golden_dir)
    enc, enc_len = torch.from_numpy(z["enc_out"]).to(DEV), torch.from_numpy(z["enc_len"]).to(DEV)
    ref, _, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=12)
    res = {}
    for mode in ("1", "0"):
        monkeypatch.setenv("ASR_AMD_DECODE_GRAPH", mode)
        model.decoder.__dict__.pop("_decode_graph", None)
        p1, l1, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=1)
        pb, lb, _ = model.decoder.batch_decode(enc[:1].contiguous(), enc_len[:1].contiguous(), max_decode_len=4)
        p0, l0, _ = model.decoder.batch_decode(enc, enc_len, max_decode_len=0)
        assert tuple(p1.shape) == (enc.shape[0], 1) and tuple(pb.shape) == (1, 4) and tuple(p0.shape) == (enc.shape[0], 0)
        np.testing.assert_array_equal(p1.cpu().numpy(), ref[:, :1].cpu().numpy())
        np.testing.assert_array_equal(pb.cpu().numpy(), ref[:1, :4].cpu().numpy())
        assert int(l0.abs().sum()) == 0
        res[mode] = (p1.cpu().numpy(), l1.cpu().numpy(), pb.cpu().numpy(), lb.cpu().numpy())
    for a, b in zip(res["1"], res["0"]):
        np.testing.assert_array_equal(a, b)


@pytest.mark.parametrize("V", [50, 1024, 1025, 4234, 4608, 5000])
def test_lsm_topk_rows_equals_the_two_kernels(V):
    g = torch.Generator().manual_seed(V)
    x = (3.0 * torch.randn(37, V, generator=g)).to(DEV)
    x[3, 7] = x[3, 11] = x[3].max() + 1.0                     # a tie for the first place: index order
    for twice in (False, True):
        v, i = ops.lsm_topk_rows(x, 5, twice=twice)
        z = ops.log_softmax_rows(x)
        v2, i2 = ops.topk_rows(ops.log_softmax_rows(z) if twice else z, 5)
        np.testing.assert_array_equal(i.cpu().numpy(), i2.cpu().numpy())
        np.testing.assert_allclose(v.cpu().numpy(), v2.cpu().numpy(), rtol=0, atol=4e-6)      # (long rows: another summation order)
        assert i[3, :2].tolist() == [7, 11]
    # a strided view (rows padded like the vocabulary projection's buffer)
    buf = torch.zeros((9, (V + 7) // 8 * 8 + 8), device=DEV)
    buf[:, :V] = x[:9]
    v3, i3 = ops.lsm_topk_rows(buf[:, :V], 3)
    np.testing.assert_array_equal(i3.cpu().numpy(), i2[:9, :3].cpu().numpy())
