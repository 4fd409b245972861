"""Checkpoint package compatibility and averaging.

The reference saves `{state_dict, optim_dict, epoch[, tr_loss, cv_loss]}` (src/transformer/transformer.py:86-97, written by
src/utils/solver.py:82-119), `optim_dict` being `torch.optim.Adam.state_dict()` of an optimizer built over `model.parameters()`
(src/transformer/train.py:166-170), resumes from it (solver.py:49-59) and averages the last N epoch files
(src/utils/average.py:33-52).  The modules here keep the reference's state_dict keys, so `state_dict` loads as it is; this module
moves the Adam moments between torch's per-parameter layout and the trainer's flat buffers and writes packages a reference
checkout can read back.

Parameter indices of `optim_dict` are positions in `model.parameters()`.  They are resolved through the package's own
`state_dict` key order (registration order = parameters() order, buffers such as `...positional_encoding.pe` skipped) and then BY
NAME into this model, so a different module registration order on this side cannot silently permute moments.
"""
import os

import torch


def _param_names_in_package_order(state_dict_keys, model):
    mine = dict(model.named_parameters())
    names = [k for k in state_dict_keys if k in mine]
    missing = [k for k in mine if k not in set(names)]
    if missing:
        raise KeyError("checkpoint lacks parameters: %s" % missing[:5])
    return names, mine


def optim_dict_from_trainer(trainer, model=None):
    """torch.optim.Adam.state_dict() of the trainer's state (indices = positions in model.parameters())."""
    model = trainer.model if model is None else model
    state = {}
    step = torch.tensor(float(trainer.step_num))
    for i, p in enumerate(model.parameters()):
        off, n = p._asr_off, p.numel()
        state[i] = {"step": step.clone(),
                    "exp_avg": trainer.m[off:off + n].detach().view(p.shape).cpu().clone(),
                    "exp_avg_sq": trainer.v[off:off + n].detach().view(p.shape).cpu().clone()}
    n_params = len(state)
    group = {"lr": trainer.lr() if trainer.step_num > 0 else 1e-3, "betas": tuple(trainer.betas), "eps": trainer.eps, "weight_decay": 0,
             "amsgrad": False, "maximize": False, "foreach": None, "capturable": False, "differentiable": False, "fused": None,
             "params": list(range(n_params))}
    return {"state": state, "param_groups": [group]}


def load_optim_dict(trainer, optim_dict, state_dict_keys=None, restore_step=True):
    """Adam moments of a reference (or own) package -> the trainer's flat m / v.  restore_step: also take the Noam step counter from
    Adam's own `step` (the reference's TransformerOptimizer does not save its counter: resumed there, the warm-up restarts at 0 -
    pass restore_step=False to reproduce that)."""
    model = trainer.model
    keys = list(state_dict_keys) if state_dict_keys is not None else list(model.state_dict().keys())
    names, mine = _param_names_in_package_order(keys, model)
    ids = optim_dict["param_groups"][0]["params"]
    if len(ids) != len(names):
        raise ValueError("optimizer has %d parameters, the model %d" % (len(ids), len(names)))
    steps = set()
    for pos, pid in enumerate(ids):
        st = optim_dict["state"].get(pid)
        p = mine[names[pos]]
        off, n = p._asr_off, p.numel()
        if st is None:                      # a parameter that never received a gradient
            trainer.m[off:off + n].zero_()
            trainer.v[off:off + n].zero_()
            continue
        if tuple(st["exp_avg"].shape) != tuple(p.shape):
            raise ValueError("optimizer state %d has shape %s, parameter %s has %s" % (pid, tuple(st["exp_avg"].shape), names[pos], tuple(p.shape)))
        trainer.m[off:off + n].copy_(st["exp_avg"].reshape(-1).to(trainer.m.device, torch.float32))
        trainer.v[off:off + n].copy_(st["exp_avg_sq"].reshape(-1).to(trainer.v.device, torch.float32))
        steps.add(int(float(st["step"])))
    if restore_step and steps:
        if len(steps) != 1:
            raise ValueError("parameters disagree on Adam's step count: %s" % sorted(steps))
        trainer.step_num = steps.pop()
    trainer._state_step = -1


def serialize(model, trainer, epoch, tr_loss=None, cv_loss=None):
    """transformer.py:86-97 - the package the reference's Solver writes with torch.save."""
    package = {"state_dict": {k: v.detach().cpu().clone() for k, v in model.state_dict().items()},
               "optim_dict": optim_dict_from_trainer(trainer, model), "epoch": epoch}
    if tr_loss is not None:
        package["tr_loss"] = tr_loss
        package["cv_loss"] = cv_loss
    return package


def load_package(package, model, trainer=None, restore_step=True):
    """`package`: a dict or a path (torch.load on CPU, like transformer.py:79-84).  Loads the weights into `model` (and, when a trainer
    is given, the Adam state into it; the trainer's bf16 weight shadow is refreshed).  -> the package dict."""
    if isinstance(package, (str, os.PathLike)):
        package = torch.load(package, map_location="cpu", weights_only=False)
    sd = package["state_dict"]
    target = model.state_dict()
    with torch.no_grad():
        for k, v in sd.items():
            if k not in target:
                raise KeyError("unexpected key in checkpoint: %s" % k)
            target[k].copy_(v.to(target[k].device, target[k].dtype))      # in place: the trainer's flat buffers stay the parameters' home
    missing = [k for k in target if k not in sd and not k.endswith("positional_encoding.pe")]
    if missing:
        raise KeyError("checkpoint lacks: %s" % missing[:5])
    if trainer is not None:
        trainer.fp.sync_shadow()
        from . import modules
        modules.bump_param_epoch()
        if "optim_dict" in package:
            load_optim_dict(trainer, package["optim_dict"], list(sd.keys()), restore_step)
    return package


def average_packages(packages):
    """utils/average.py:40-48 - the FIRST package with its state_dict replaced by the mean of all state_dicts (everything else,
    optimizer state included, is the first package's)."""
    out = dict(packages[0])
    n = len(packages)
    avg = {k: torch.zeros_like(v) for k, v in packages[0]["state_dict"].items()}
    for pkg in packages:
        for k in avg:
            avg[k] += pkg["state_dict"][k] / n
    out["state_dict"] = avg
    return out


def average_last(expdir, num, save=True):
    """utils/average.py:33-52: the `num` last files named epoch*.model in `expdir` - last by the reference's plain string sort
    (so epoch-9 ranks above epoch-10: its quirk, kept) - averaged and saved as avg-last<num>.model."""
    names = sorted(t for t in os.listdir(expdir) if t.startswith("epoch") and t.endswith(".model"))
    names.reverse()
    names = names[:num]
    if not names:
        raise FileNotFoundError("no epoch*.model files in %s" % expdir)
    pkg = average_packages([torch.load(os.path.join(expdir, fn), map_location="cpu", weights_only=False) for fn in names])
    if save:
        torch.save(pkg, os.path.join(expdir, "avg-last%d.model" % len(names)))
    return pkg, names
