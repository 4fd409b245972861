"""Host-side logic that needs no GPU: state_dict contract, target preprocessing, mask helpers, LR schedule."""
import argparse
import os

import numpy as np
import pytest
import torch

import asr_amd
from oracle import asr_oracle as O
from weights import names_shapes_from_json


def _args(z):
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    return argparse.Namespace(spec_aug_cfg=None, **cfg)


def test_state_dict_contract_conv_ctc_transformer(golden_dir):
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    model = asr_amd.Conv_CTC_Transformer.create_model(_args(z))
    ours = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert sorted(ours) == sorted(names_shapes_from_json(z["names_shapes"]))


def test_state_dict_contract_cif_model(golden_dir):
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    model = asr_amd.CIF_Model.create_model(_args(z))
    ours = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert sorted(ours) == sorted(names_shapes_from_json(z["names_shapes"]))


def test_state_dict_contract_ctc_model(golden_dir):
    from asr_amd.ctc_model import CTC_Model, Decoder, Encoder
    z = np.load(os.path.join(golden_dir, "g5_ctc_model.npz"))
    model = CTC_Model(Encoder(80, 2, 2, 64, 64, 64, 128, dropout=0.0, pe_maxlen=5000), Decoder(50, 64))
    ours = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert sorted(ours) == sorted(names_shapes_from_json(z["names_shapes"]))


def test_state_dict_contract_ctc_transformer(golden_dir):
    z = np.load(os.path.join(golden_dir, "g1_ctc_transformer.npz"))
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0),
                                    asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
    ours = [(k, tuple(v.shape)) for k, v in model.state_dict().items()]
    assert sorted(ours) == sorted(names_shapes_from_json(z["names_shapes"]))


def test_positional_encoding_buffer_matches_reference(golden_dir):
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    pe = asr_amd.PositionalEncoding(64).pe[0, :128].numpy()
    np.testing.assert_allclose(pe, z["pe_head"], rtol=0, atol=1e-6)


def test_decoder_preprocess_matches_reference_and_oracle(golden_dir):
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))
    dec = asr_amd.Decoder(2, 3, 50, 1, 2, 64, 128, dropout=0.0)
    tg = torch.from_numpy(z["targets"])
    ys_in, ys_out = dec.preprocess(tg)
    o_in, o_out = O.decoder_preprocess(z["targets"], 2, 3)
    np.testing.assert_array_equal(ys_out.numpy(), z["targets_eos"])
    np.testing.assert_array_equal(ys_in.numpy(), o_in)
    np.testing.assert_array_equal(ys_out.numpy(), o_out)
    # zeros in the middle are stripped like decoder.py:46 does
    tg2 = torch.tensor([[5, 0, 6, 7], [0, 0, 9, 0]])
    i2, o2 = dec.preprocess(tg2)
    oi, oo = O.decoder_preprocess(tg2.numpy(), 2, 3)
    np.testing.assert_array_equal(i2.numpy(), oi)
    np.testing.assert_array_equal(o2.numpy(), oo)


def test_mask_helpers_match_oracle():
    lens = torch.tensor([5, 3, 1])
    np.testing.assert_array_equal(asr_amd.utils.sequence_mask(lens).numpy(), O.sequence_mask(lens.numpy()))
    np.testing.assert_array_equal(asr_amd.utils.get_attn_pad_mask(lens, 4).numpy(), O.get_attn_pad_mask(lens.numpy(), 4))
    seq = torch.tensor([[2, 7, 0], [2, 0, 0]])
    np.testing.assert_array_equal(asr_amd.utils.get_subsequent_mask(seq).numpy(), O.get_subsequent_mask(seq.numpy()))
    np.testing.assert_array_equal(asr_amd.utils.get_attn_key_pad_mask(seq, seq, 0).numpy(),
                                  O.get_attn_key_pad_mask(seq.numpy(), seq.numpy(), 0))


def test_conv_length_rule():
    # ceil(len/2) per layer as int32 (conv_encoder.py:113-115)
    conv = asr_amd.Conv2dSubsample(80, 64, n_layers=2)
    lens = torch.tensor([100, 90, 77, 64, 1])
    exp = np.ceil(np.ceil(lens.numpy() / 2.0) / 2.0).astype(np.int32)
    got = lens.clone()
    for _ in range(2):
        got = torch.div(got + 1, 2, rounding_mode="floor")
    np.testing.assert_array_equal(got.numpy(), exp)
    assert conv.d_conv_out == 40


def test_dropout_keys_match_the_oracle_restatement():
    from oracle import asr_oracle as O
    for seed, name, call in ((0, "encoder.dropout", 1), (606, "decoder.layer_stack.1.enc_attn.attention.dropout", 3), (2**63 + 5, "x", 10**6)):
        assert asr_amd.dropout_site_keys(seed, name, call) == O.dropout_site_keys(seed, name, call)
    assert asr_amd.dropout_thr16(0.1) == 6554 and asr_amd.dropout_thr16(0.0) == 0


@pytest.mark.parametrize("gen,inc", [("gen_attn_fwd4.py", "attention_fwd4_asm.inc"), ("gen_attn_bwd.py", "attention_bwd_asm.inc"),
                                     ("gen_attn_bwd_dq.py", "attention_bwd_dq_asm.inc"), ("gen_ffn_fwd.py", "ffn_fwd2_asm.inc")])
def test_generated_attention_streams_are_current(gen, inc):
    """csrc/attention_*_asm.inc and csrc/ffn_fwd2_asm.inc (+ ffn_fwd2_params.h) are the output of tools/gen_*.py: an edit of one without the
    other must not go unnoticed."""
    import subprocess, sys, tempfile
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    with tempfile.TemporaryDirectory() as d:
        out = os.path.join(d, "x.inc")
        subprocess.run([sys.executable, os.path.join(root, "tools", gen), "--out", out], check=True, stderr=subprocess.DEVNULL)
        assert open(out).read() == open(os.path.join(root, "end-to-end_asr_pytorch_amd", "csrc", inc)).read()
        if gen == "gen_ffn_fwd.py":
            assert open(os.path.join(d, "ffn_fwd2_params.h")).read() == open(os.path.join(root, "end-to-end_asr_pytorch_amd", "csrc", "ffn_fwd2_params.h")).read()
