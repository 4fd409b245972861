"""Checkpoint package compatibility on the CPU: the reference's own package after two optimizer steps
(tests/golden/g8_checkpoint.npz, written by the reference's serialize()) goes into the flat optimizer layout by name, comes back out
as a torch.optim.Adam state_dict that torch itself accepts, and averaging follows src/utils/average.py."""
import os

import numpy as np
import torch

import asr_amd
from asr_amd import checkpoint
from asr_amd.trainer import _param_order, flat_offsets


class FlatState:
    """The part of Trainer the checkpoint code touches, on CPU tensors (Trainer itself needs the GPU for its bf16 shadow)."""

    def __init__(self, model):
        self.model = model
        order = _param_order(model)
        offs, n = flat_offsets(order)
        for p, off in zip(order, offs):
            p._asr_off = off
        self.m, self.v = torch.zeros(n), torch.zeros(n)
        self.step_num, self.betas, self.eps, self._state_step = 0, (0.9, 0.98), 1e-9, 0
        self.k, self.init_lr, self.warmup = 0.2, model.encoder.d_model ** -0.5, 4000

    def lr(self):
        return self.k * self.init_lr * min(self.step_num ** (-0.5), self.step_num * self.warmup ** (-1.5))


def load_fixture(golden_dir):
    z = np.load(os.path.join(golden_dir, "g8_checkpoint.npz"))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    keys = str(z["sd_keys"]).split("|")
    sd = {k: torch.from_numpy(z["sd:" + k]) for k in keys if ("sd:" + k) in z.files}
    n = len(z["opt_params"])
    optim = {"state": {i: {"step": torch.tensor(float(z["opt:%d:step" % i])), "exp_avg": torch.from_numpy(z["opt:%d:exp_avg" % i]),
                           "exp_avg_sq": torch.from_numpy(z["opt:%d:exp_avg_sq" % i])} for i in range(n)},
             "param_groups": [{"lr": float(z["opt_lr"]), "betas": tuple(z["opt_betas"].tolist()), "eps": float(z["opt_eps"]),
                               "weight_decay": float(z["opt_weight_decay"]), "amsgrad": False, "params": z["opt_params"].tolist()}]}
    pe_keys = [k for k in keys if k.endswith("positional_encoding.pe")]
    package = {"state_dict": sd, "optim_dict": optim, "epoch": int(z["epoch"]), "tr_loss": torch.from_numpy(z["tr_loss"]),
               "cv_loss": torch.from_numpy(z["cv_loss"])}
    return z, cfg, package, keys, pe_keys


def build(cfg):
    enc = asr_amd.Encoder(cfg["d_input"], cfg["n_layers_enc"], cfg["n_head"], cfg["d_model"], cfg["d_inner"], dropout=0.0)
    dec = asr_amd.Decoder(cfg["sos_id"], cfg["eos_id"], cfg["vocab_size"], cfg["n_layers_dec"], cfg["n_head"], cfg["d_model"], cfg["d_inner"],
                          dropout=0.0)
    return asr_amd.CTC_Transformer(enc, dec)


def with_pe(package, model, pe_keys):
    """the fixture does not store the (deterministic) positional-encoding buffers the reference's state_dict carries"""
    sd = dict(package["state_dict"])
    for k in pe_keys:
        sd[k] = model.state_dict()[k].clone()
    return dict(package, state_dict={k: sd[k] for k in list(model.state_dict().keys())})


def test_reference_package_loads_by_name_and_round_trips_through_torch_adam(golden_dir):
    z, cfg, package, keys, pe_keys = load_fixture(golden_dir)
    model = build(cfg)
    # same key set and order as the reference's state_dict
    assert list(model.state_dict().keys()) == keys
    st = FlatState(model)
    pkg = with_pe(package, model, pe_keys)
    checkpoint.load_package(pkg, model)
    checkpoint.load_optim_dict(st, pkg["optim_dict"], list(pkg["state_dict"].keys()))
    assert st.step_num == 2
    names = [k for k in keys if not k.endswith(".pe")]
    mine = dict(model.named_parameters())
    for i, name in enumerate(names):
        p = mine[name]
        np.testing.assert_array_equal(p.detach().numpy(), z["sd:" + name])
        off, n = p._asr_off, p.numel()
        np.testing.assert_array_equal(st.m[off:off + n].view(p.shape).numpy(), z["opt:%d:exp_avg" % i])
        np.testing.assert_array_equal(st.v[off:off + n].view(p.shape).numpy(), z["opt:%d:exp_avg_sq" % i])
    # and back: what serialize() writes is a state_dict torch's own Adam loads (the reference's resume path, solver.py:49-59)
    out = checkpoint.serialize(model, st, epoch=2, tr_loss=package["tr_loss"], cv_loss=package["cv_loss"])
    assert set(out) == {"state_dict", "optim_dict", "epoch", "tr_loss", "cv_loss"} and out["epoch"] == 2
    assert list(out["state_dict"].keys()) == keys
    adam = torch.optim.Adam(model.parameters(), betas=(0.9, 0.98), eps=1e-9)
    adam.load_state_dict(out["optim_dict"])
    for i, p in enumerate(model.parameters()):
        s = adam.state[p]
        np.testing.assert_array_equal(s["exp_avg"].numpy(), z["opt:%d:exp_avg" % i])
        np.testing.assert_array_equal(s["exp_avg_sq"].numpy(), z["opt:%d:exp_avg_sq" % i])
        assert float(s["step"]) == 2.0
    np.testing.assert_allclose(out["optim_dict"]["param_groups"][0]["lr"], float(z["opt_lr"]), rtol=1e-12)


def test_resumed_torch_adam_reproduces_the_references_third_step(golden_dir):
    """the package's state is complete: a stock torch Adam resumed from OUR serialize() output, fed the reference-equivalent CPU
    gradient (oracle/torch_cpu_ref.py), lands on the reference's parameters after its third step"""
    from oracle import torch_cpu_ref as R
    z, cfg, package, keys, pe_keys = load_fixture(golden_dir)
    model = build(cfg)
    st = FlatState(model)
    pkg = with_pe(package, model, pe_keys)
    checkpoint.load_package(pkg, model)
    checkpoint.load_optim_dict(st, pkg["optim_dict"], list(pkg["state_dict"].keys()))
    out = checkpoint.serialize(model, st, epoch=2)
    leaves = R.leaves(out["state_dict"])
    names = [k for k in keys if not k.endswith(".pe")]
    params = [leaves[k] for k in names]
    adam = torch.optim.Adam(params, betas=(0.9, 0.98), eps=1e-9)
    adam.load_state_dict(out["optim_dict"])
    R.joint_step(leaves, torch.from_numpy(z["x"]), torch.from_numpy(z["lens"]), torch.from_numpy(z["targets"]), cfg, p=0.0, train=False)
    st.step_num += 1
    for g in adam.param_groups:
        g["lr"] = st.lr()
    np.testing.assert_allclose(st.lr(), float(z["lr_step3"]), rtol=1e-12)
    adam.step()
    for k in names:
        np.testing.assert_allclose(leaves[k].detach().numpy(), z["after3:" + k], rtol=1e-5, atol=2e-7, err_msg=k)


def test_average_follows_the_references_script(tmp_path, golden_dir):
    z, cfg, package, keys, pe_keys = load_fixture(golden_dir)
    model = build(cfg)
    base = with_pe(package, model, pe_keys)
    for ep, scale in ((8, 1.0), (9, 2.0), (10, 4.0), (11, 8.0)):
        pkg = dict(base, epoch=ep, state_dict={k: v * scale for k, v in base["state_dict"].items()})
        torch.save(pkg, os.path.join(tmp_path, "epoch-%d.model" % ep))
    torch.save(base, os.path.join(tmp_path, "final.model"))      # not an epoch file: ignored
    avg, used = checkpoint.average_last(str(tmp_path), 2)
    # plain string sort, reversed (average.py:35-38): "epoch-9" and "epoch-8" outrank "epoch-11" / "epoch-10"
    assert used == ["epoch-9.model", "epoch-8.model"]
    k = "ctc_fc.weight"
    np.testing.assert_allclose(avg["state_dict"][k].numpy(), base["state_dict"][k].numpy() * 1.5, rtol=1e-6)
    assert avg["epoch"] == 9                                        # everything but state_dict is the first package's
    re = torch.load(os.path.join(tmp_path, "avg-last2.model"), map_location="cpu", weights_only=False)
    np.testing.assert_array_equal(re["state_dict"][k].numpy(), avg["state_dict"][k].numpy())
