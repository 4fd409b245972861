"""Loss functions with the reference's names and argument meaning (src/transformer/loss.py, src/ctcModel/loss.py),
computed by the fused HIP kernels (ctc.hip, ce.hip).  Each returns 0-dim tensors like the reference.
Gradients flow through torch.autograd.Function wrappers whose backward is the fused HIP gradient kernel.
"""
import torch

from . import ops


class _CtcLossFn(torch.autograd.Function):
    """F.log_softmax + F.ctc_loss(blank=V-1, reduction='mean', zero_infinity=False) (loss.py:41-43)."""

    @staticmethod
    def forward(ctx, logits, in_len, targets):
        loss, nll, st = ops.ctc_loss_fwd(logits.detach(), in_len, targets)
        ctx.st = st
        ctx.mark_non_differentiable(nll)
        return loss.reshape(()), nll

    @staticmethod
    def backward(ctx, gout, _gnll):
        return ops.ctc_loss_bwd(ctx.st, gout), None, None


class _CeLossFn(torch.autograd.Function):
    """Label-smoothed CE (loss.py:5-31)."""

    @staticmethod
    def forward(ctx, logits2d, targets1d, smoothing):
        loss2, row_loss, lse, tg = ops.ce_loss_fwd(logits2d.detach(), targets1d, smoothing)
        ctx.save = (logits2d.detach(), tg, float(smoothing), lse, loss2)
        return loss2[0].reshape(())

    @staticmethod
    def backward(ctx, gout):
        logits2d, tg, smoothing, lse, loss2 = ctx.save
        return ops.ce_loss_bwd(logits2d, tg, smoothing, lse, loss2, gout), None, None


def ctc_loss(logits, len_logits, targets):
    """-> (mean loss, per-sample nll)"""
    if logits.stride(-1) != 1 or logits.stride(0) != logits.shape[1] * logits.stride(1):
        logits = logits.contiguous()
    return _CtcLossFn.apply(logits, len_logits, targets)


def cal_ce_loss(logits, targets, smoothing=0.0):
    """src/transformer/loss.py:5-31."""
    V = logits.size(-1)
    logits2d = logits.reshape(-1, V)
    if logits2d.stride(1) != 1:
        logits2d = logits2d.contiguous()
    return _CeLossFn.apply(logits2d, targets.contiguous().view(-1), smoothing)


def cal_ctc_ce_loss(logits_ctc, len_logits_ctc, logits_ce, targets, smoothing=0.0):
    """src/transformer/loss.py:34-48 — CTC targets are `targets` as given (the solver passes targets_eos)."""
    ctc, _ = ctc_loss(logits_ctc, len_logits_ctc, targets)
    return ctc, cal_ce_loss(logits_ce, targets, smoothing)


def cal_ctc_qua_ce_loss(logits_ctc, len_logits_ctc, _number, number, logits_ce, targets, smoothing=0.0):
    """src/transformer/loss.py:51-61."""
    qua_loss = torch.pow(_number - number, 2).mean()
    ctc, ce = cal_ctc_ce_loss(logits_ctc, len_logits_ctc, logits_ce, targets, smoothing)
    return qua_loss, ctc, ce


def cal_loss(logits, len_logits, gold, smoothing=0.0):
    """src/ctcModel/loss.py:4-13."""
    return ctc_loss(logits, len_logits, gold)[0]
