"""Host-side trainer logic that needs no GPU: parameter ordering / adjacency, bucket cover, Noam schedule."""
import numpy as np
import torch

import asr_amd
from asr_amd.trainer import _param_order, flat_offsets
from oracle import asr_oracle as O


def _model():
    return asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))


def test_param_order_is_a_permutation_with_qkv_adjacent():
    m = _model()
    order = _param_order(m)
    assert sorted(id(p) for p in order) == sorted(id(p) for p in m.parameters())
    pos = {id(p): i for i, p in enumerate(order)}
    cross = {id(layer.enc_attn) for mod in m.modules() if isinstance(mod, asr_amd.Decoder) for layer in mod.layer_stack}
    selfs = [mod for mod in m.modules() if isinstance(mod, asr_amd.MultiheadAttention) and id(mod) not in cross]
    assert selfs and cross
    for mod in selfs:
        i = pos[id(mod.w_qs.weight)]
        assert pos[id(mod.w_ks.weight)] == i + 1 and pos[id(mod.w_vs.weight)] == i + 2
        j = pos[id(mod.w_qs.bias)]
        assert pos[id(mod.w_ks.bias)] == j + 1 and pos[id(mod.w_vs.bias)] == j + 2
    offs, total = flat_offsets(order)
    assert all(o % 8 == 0 for o in offs) and total >= sum(p.numel() for p in order)
    # Q/K/V stay adjacent (no padding inside the triples) so their concatenation is a plain view
    for mod in selfs:
        i = pos[id(mod.w_qs.weight)]
        assert offs[i + 1] == offs[i] + mod.w_qs.weight.numel() and offs[i + 2] == offs[i + 1] + mod.w_ks.weight.numel()
    # the decoder's cross-attention K/V weights (then biases) of ALL layers form one contiguous run (Decoder._cross_kv)
    for mod in m.modules():
        if isinstance(mod, asr_amd.Decoder):
            for group in mod.cross_kv_params():
                i0 = pos[id(group[0])]
                assert [pos[id(p)] for p in group] == list(range(i0, i0 + len(group)))
                assert all(offs[i0 + k + 1] == offs[i0 + k] + group[k].numel() for k in range(len(group) - 1))


def test_flat_offsets_survive_odd_sized_parameters():
    import argparse
    args = argparse.Namespace(d_input=80, LFR_m=1, d_model=64, n_conv_layers=2, n_layers_enc=1, n_head=2, d_inner=128, dropout=0.0,
                              sos_id=2, eos_id=3, vocab_size=50, n_layers_dec=1, spec_aug_cfg=None, d_assigner_hidden=32, w_context=3,
                              n_assigner_layers=2)
    m = asr_amd.CIF_Model.create_model(args)      # has a 1-element parameter (assigner.linear.bias)
    order = _param_order(m)
    offs, _ = flat_offsets(order)
    assert any(p.numel() == 1 for p in order)
    assert all(o % 8 == 0 for o in offs)


def test_noam_schedule_matches_oracle_and_reference_pins(golden_dir):
    import os
    z = np.load(os.path.join(golden_dir, "g0_conv_ctc_transformer.npz"))

    class T:  # the trainer's lr() without building device buffers
        k, warmup, init_lr = 0.2, 4000, 256 ** (-0.5)
        lr = asr_amd.Trainer.lr
    t = T()
    got = []
    for n in (1, 4000, 10000):
        t.step_num = n
        got.append(t.lr())
        np.testing.assert_allclose(t.lr(), O.noam_lr(n, 0.2, 256, 4000), rtol=1e-12)
    np.testing.assert_allclose(got, z["noam_k0.2_d256_w4000_steps_1_4000_10000"], rtol=1e-12)
