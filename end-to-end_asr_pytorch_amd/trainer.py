"""Data-parallel training step for the hot path.

Mirrors what the reference's harness does per batch (src/transformer/solver.py:83-93, optimizer.py:19-29,
train.py:166-170) with every piece on the GPU path:

    forward (HIP kernels, backward tape recorded) -> loss = ctc + ce (fused CTC / CE kernels) -> backward (tape: HIP
    kernels write parameter gradients straight into one flat fp32 buffer) -> RCCL all-reduce of that buffer in
    buckets, launched as soon as a bucket's last gradient is final so it overlaps the rest of the backward ->
    Noam learning rate -> fused Adam over the flat buffer (which also refreshes the bf16 MFMA weight shadow).

One process per GPU; utterances are sharded across ranks (each rank owns B_local of them, the same number on every rank);
gradients are summed by the all-reduce and 1/world is folded into the Adam kernel's grad_scale.  The result is the gradient of
the GLOBAL-batch loss, not DDP's mean of per-rank means: CTC ('mean' = mean over utterances of nll / target length,
loss.py:41-43) and the CIF quantity loss (mean over utterances) are already rank-separable with equal B_local, and the CE
denominator n_word (loss.py:22-25) is all-reduced so that rank r seeds its CE backward with world * n_word_r / n_word_global
(`exact_global_mean`; one 4-byte all-reduce per step).  tests/test_gpu_dp.py checks 2 ranks x B/2 against 1 rank x B.
"""
import math
import os

import torch

from . import modules, ops
from .ctc_model import CTC_Model
from .loss import cal_ce_loss  # noqa: F401  (API parity)


def _param_order(model):
    """Parameters in forward-execution module order with each attention block's Q,K,V weights (then biases) adjacent, so
    their concatenations are plain views of the flat buffers."""
    seen, order = set(), []

    def add(p):
        if id(p) not in seen:
            seen.add(id(p))
            order.append(p)

    for m in model.modules():
        if isinstance(m, modules.Decoder):       # cross-attention K/V weights (biases) of all layers adjacent: Decoder._cross_kv
            ws, bs = m.cross_kv_params()
            for p in ws + bs:
                add(p)
        if isinstance(m, modules.MultiheadAttention):
            for p in (m.w_qs.weight, m.w_ks.weight, m.w_vs.weight, m.w_qs.bias, m.w_ks.bias, m.w_vs.bias):
                add(p)
        for p in m.parameters(recurse=False):
            add(p)
    for p in model.parameters():
        add(p)
    return order


def flat_offsets(order):
    """Element offsets of the parameters in one flat buffer: each starts on a multiple of 8 elements (16 bytes of bf16), so odd-sized
    parameters (e.g. a 1-element bias) cannot misalign the MFMA operands that follow; Q/K/V weight (bias) triples stay adjacent
    because their sizes are multiples of 64.  Returns (offsets, total)."""
    offs, n = [], 0
    for p in order:
        offs.append(n)
        n = (n + p.numel() + 7) // 8 * 8
    return offs, n


class FlatParams:
    """Re-homes every parameter into one flat fp32 buffer, with a flat gradient buffer and a flat bf16 shadow."""

    def __init__(self, model, device):
        self.params = _param_order(model)
        offs, n = flat_offsets(self.params)
        self.numel = n
        total = (n + 63) // 64 * 64
        self.flat = torch.zeros(total, device=device, dtype=torch.float32)
        self.grad = torch.zeros(total, device=device, dtype=torch.float32)
        self.flat16 = torch.zeros(total, device=device, dtype=torch.bfloat16)
        self.offsets = offs
        for p, off in zip(self.params, offs):
            k = p.numel()
            self.flat[off:off + k].copy_(p.data.reshape(-1).to(device))
            p.data = self.flat[off:off + k].view(p.shape)
            p.grad = self.grad[off:off + k].view(p.shape)
            p._asr_off, p._asr_gflat, p._asr_flat16, p._asr_flat32 = off, self.grad, self.flat16, self.flat
        self.sync_shadow()

    def sync_shadow(self):
        """bf16 shadow <- fp32 master (after init / load_state_dict; the Adam kernel keeps it fresh afterwards)."""
        self.flat16.copy_(ops.cast_bf16(self.flat))

    def check_alignment(self):
        for p, off in zip(self.params, self.offsets):
            if p.dim() >= 2 and off % 8 != 0:
                raise RuntimeError("parameter of shape %s lands at flat offset %d (not 16-byte aligned for bf16 MFMA loads)"
                                   % (tuple(p.shape), off))


GRAPH_EXEC = True      # the captured step launched by the multi-stream executor (False: hipGraphLaunch, which runs a graph's branches level by level: +1.5-2.5 ms)


class GradBuckets:
    """Bucketed gradient all-reduce over one flat buffer, overlapped with the backward.

    Buckets are contiguous ranges of the flat gradient cut at parameter boundaries.  `on_done(params)` is called after each
    backward closure with the parameters whose gradients just became final; when a bucket's last parameter is done its
    all-reduce (sum) is launched asynchronously (RCCL on the GPU, gloo in the CPU tests) while the backward keeps running.
    `finish()` waits for all of them.  Device-agnostic: only torch.distributed + tensor views."""

    def __init__(self, flat_grad, params, offsets, numel, n_buckets=4, group=None):
        self.flat_grad, self.group = flat_grad, group
        self.launch_stream = None      # when set: all-reduces are issued from this stream, after an event of the current one
        self.mark_stream = None        # when set (a step being captured for the graph executor): a bucket's ready point is a marker node
                                       # ... on this stream instead of an all-reduce
        self.world = 1
        if torch.distributed.is_available() and torch.distributed.is_initialized():
            self.world = torch.distributed.get_world_size(group)
        cuts = [0]
        for i in range(1, n_buckets):
            target = numel * i // n_buckets
            off = min(offsets, key=lambda o: abs(o - target))
            if off > cuts[-1]:
                cuts.append(off)
        cuts.append(flat_grad.numel())
        self.ranges = [(cuts[i], cuts[i + 1]) for i in range(len(cuts) - 1)]
        self._bucket_of, self._size = {}, [0] * len(self.ranges)
        for p, off in zip(params, offsets):
            for bi, (a, b) in enumerate(self.ranges):
                if a <= off < b:
                    self._bucket_of[id(p)] = bi
                    self._size[bi] += 1
        self.start()

    def start(self):
        self._pending, self._handles, self.launch_order = list(self._size), [], []

    def on_done(self, params):
        for p in params:
            bi = self._bucket_of[id(p)]
            self._pending[bi] -= 1
            if self._pending[bi] == 0:
                self.launch_order.append(bi)
                modules.flush_wgrads()                     # (decoder-sized weight gradients still waiting for their grouped launch)
                if self.mark_stream is not None:
                    # captured step: the collective is a node of its own chain (the comm stream), ordered after every stream that
                    # wrote this bucket; the executor's C loop makes the RCCL call there (ops.collective_mark)
                    a, b = self.ranges[bi]
                    cur = torch.cuda.current_stream()
                    self.mark_stream.wait_stream(cur)
                    if self.launch_stream is not None:
                        self.mark_stream.wait_stream(self.launch_stream)
                    with torch.cuda.stream(self.mark_stream):
                        ops.collective_mark(self.flat_grad[a:b], tag=bi)
                elif self.world > 1:
                    a, b = self.ranges[bi]
                    if self.launch_stream is not None:     # gradients of this bucket are written on two streams: order after both
                        ev = torch.cuda.Event()
                        ev.record(torch.cuda.current_stream())
                        self.launch_stream.wait_event(ev)
                        with torch.cuda.stream(self.launch_stream):
                            self._handles.append(torch.distributed.all_reduce(self.flat_grad[a:b], group=self.group, async_op=True))
                    else:
                        self._handles.append(torch.distributed.all_reduce(self.flat_grad[a:b], group=self.group, async_op=True))

    def finish(self):
        if self.mark_stream is not None:
            torch.cuda.current_stream().wait_stream(self.mark_stream)
        for h in self._handles:
            h.wait()
        if any(c != 0 for c in self._pending):
            raise RuntimeError("gradient bucket(s) never became ready: pending=%s" % self._pending)


class Trainer:
    """model: asr_amd.CTC_Transformer (or Transformer-family module that records a tape).  k, warmup: Noam schedule
    (optimizer.py:24-29); betas / eps as configured at train.py:166-170."""

    def __init__(self, model, k=0.2, warmup_steps=4000, betas=(0.9, 0.98), eps=1e-9, label_smoothing=0.1, n_buckets=8,
                 process_group=None, lambda_qua=0.001, overlap_ctc=None, exact_global_mean=True, force_collective=None):
        self.model = model
        self.group = process_group
        self.exact_global_mean = exact_global_mean
        # collective nodes in the captured step even with one rank (a 1-rank RCCL communicator: the executor's all-reduce plumbing,
        # measurable on a 1-GPU box; tools/rccl_sanity.py, tests/test_gpu_graph.py)
        self.force_collective = bool(force_collective)
        # CTC branch on a side stream beside the decoder branch (see _ctc_side_branch); Trainer(overlap_ctc=False) serialises it
        self.overlap_ctc = True if overlap_ctc is None else bool(overlap_ctc)
        # (Measured and not kept, round 5: the branch queued BEHIND the decoder's first launch - the K / V projection of all six layers over
        # the encoder output, a chip-filling GEMM that shares the chip with the branch's own projection - decoder segment 2.32 -> 2.24-2.39 ms,
        # step unchanged within the box's noise; that projection split into layer 0 on the launch stream and layers 1-5 on a side stream:
        # segment + 0.12 ms, the small GEMM waits for the CUs the branch's projection holds.)
        # launch budget of the branch's backward (ops.launch_budget: CTC gradient pass 3 x 128 workgroups, ctc_fc's two GEMMs 128 each):
        # at full grids they starve the decoder's fat-workgroup kernels for 100-150 us apiece; 0 = none (tools/op_timeline.py)
        self.side_budget = 128
        self.wgrad_stream = True     # weight-gradient GEMMs on a side stream (backward()); bench.py's kernel-alone pass sets it False
        self.lambda_qua = lambda_qua      # CIF models: loss = lambda_qua * qua + ctc + ce (solver.py:153, train.py:64)
        dev = next(model.parameters()).device
        self.fp = FlatParams(model, dev)
        self.fp.check_alignment()
        self.m = torch.zeros_like(self.fp.flat)
        self.v = torch.zeros_like(self.fp.flat)
        self.k, self.warmup, self.betas, self.eps = k, warmup_steps, betas, eps
        self.init_lr = model.encoder.d_model ** (-0.5)
        self.step_num = 0
        self.smoothing = label_smoothing
        self.buckets = GradBuckets(self.fp.grad, self.fp.params, self.fp.offsets, self.fp.numel, n_buckets, process_group)
        self.world = self.buckets.world
        if self.world > 1:
            # every rank starts from rank 0's parameters (what DDP's constructor does): ranks built with different seeds would
            # otherwise all-reduce gradients onto different weights and diverge silently
            src = torch.distributed.get_global_rank(process_group, 0) if process_group is not None else 0
            torch.distributed.broadcast(self.fp.flat, src=src, group=process_group)
            self.fp.sync_shadow()
        self._graph, self._graph_key, self._graph_out, self._graph_failed, self._eager_steps = None, None, None, None, 0
        self._nw_marked = None
        self._graph_in = None      # the input buffers the graph was captured against (step_graphed copies each batch into them)
        self._graphx = None        # ops.GraphExec over the captured graph (multi-stream launch of its nodes), when available
        self._state, self._state_step = None, -1
        self.launch_mode, self.launch_timing = None, None      # step_auto's choice ("eager" | "graph") and what it measured
        self._nw_handle = None
        self._hook_shadow_sync()

    def _hook_shadow_sync(self):
        """model.load_state_dict() after construction rewrites the fp32 master in place: refresh the bf16 shadow the kernels read."""
        fp = self.fp

        def post(module, incompatible):
            fp.sync_shadow()
            modules.bump_param_epoch()
        try:
            self.model.register_load_state_dict_post_hook(post)
        except AttributeError:      # very old torch: callers use Trainer.load_state_dict
            pass

    # ---- checkpoint state (the reference's package: transformer.py:86-97 keeps model.state_dict() + optimizer.state_dict()) ----
    def state_dict(self):
        return {"step_num": self.step_num, "m": self.m.detach().clone(), "v": self.v.detach().clone(), "k": self.k,
                "warmup_steps": self.warmup, "init_lr": self.init_lr}

    def load_state_dict(self, sd):
        self.step_num = int(sd["step_num"])
        self.m.copy_(sd["m"].to(self.m.device))
        self.v.copy_(sd["v"].to(self.v.device))
        self._state_step = -1
        self.fp.sync_shadow()
        modules.bump_param_epoch()

    def lr(self):
        """optimizer.py:24-29 (step_num already incremented)."""
        return self.k * self.init_lr * min(self.step_num ** (-0.5), self.step_num * (self.warmup ** (-1.5)))

    def _ctc_side_branch(self, enc, lens, ctc_targets, pre_event=None):
        """The CTC branch of the joint models - ctc_fc projection, CTC loss forward AND backward, ctc_fc's own backward - depends
        on nothing but the encoder output, while the decoder branch next to it is a long run of small kernels (M = B*(U+1) rows)
        that leave most of the chip idle.  So it is queued on a side stream right after the encoder forward; its gradient wrt
        the encoder output lands in a proxy activation that a `join` closure - placed on the main tape exactly where the
        encoder's backward begins - adds in.  Called from the model's forward through `_ctc_hook`."""
        model = self.model
        main = torch.cuda.current_stream()
        # (side_inline: the same launches in the same order on the launch stream itself - bench.py's kernel-alone timing pass)
        aux = main if getattr(self, "side_inline", False) else ops.aux_stream(enc.f32.device)
        if aux is not main:
            aux.wait_stream(main)
        if pre_event is not None:
            aux.wait_event(pre_event)
            ctc_targets.record_stream(aux)
        proxy = modules.Act(enc.f32, enc.b16, enc.B, enc.L)
        with torch.cuda.stream(aux):
            with modules.record() as side_tape:
                logits, done = modules._vocab_proj(model, "ctc", model.ctc_fc.weight, proxy, ctc=(ctc_targets, lens))
            if done is not None:      # projection + table rows in one launch, the recursion has run on the table (ops.vocab_proj_ctc)
                ctc, nll, st = done
            else:
                ctc, nll, st = ops.ctc_loss_fwd(logits.view(enc.B, enc.L, -1), ops.as_i32(lens, logits.device), ctc_targets)
            with ops.launch_budget(self.side_budget if aux is not main else 0):
                model._grad_slots["ctc"]["g"] = ops.ctc_loss_bwd(st, torch.ones(1, device=logits.device), bf16=(modules.get_precision() == "bf16"))
                side_tape.backward()
        self._side = {"ctc": ctc, "st": st}
        params = (model.ctc_fc.weight,)

        seg = self.__dict__.get("_seg_events")      # tools/step_segments.py: GPU time of the decoder segment (encoder forward's end ->
        if seg is not None:                          # ... the point where the encoder's backward begins)
            e = torch.cuda.Event(enable_timing=True)
            e.record(main)
            seg.append(("enc_fwd_end", e))

        def join():
            if seg is not None:
                e2 = torch.cuda.Event(enable_timing=True)
                e2.record(torch.cuda.current_stream())
                seg.append(("enc_bwd_begin", e2))
            take()

        def take():
            """the side branch's gradient wrt the encoder output joins enc.grad: here at the latest, or earlier from whoever adds the
            next contribution (Decoder._cross_kv's data-gradient GEMM takes enc.grad as its addend: no add launch of its own over
            the [B*L, d_model] gradient)"""
            if enc.lazy_grad is None:
                return
            enc.lazy_grad = None
            main_now = torch.cuda.current_stream()
            if aux is not main_now:
                main_now.wait_stream(aux)
            g = proxy.grad
            g.record_stream(main_now)               # allocated on the side stream, last read here
            modules._acc(enc, g)
            proxy.grad = None

        enc.lazy_grad = take
        modules._TAPE.push(join, params)
        ctc.record_stream(main)
        return logits

    def forward_loss(self, feats, lens, targets, noise=None, max_target_len=None):
        """forward + joint loss with the tape recorded; returns (ctc, ce, state for backward).  max_target_len: the longest
        target (non-pad tokens) of the batch if the caller knows it - the step then has no host synchronisation at all."""
        model = self.model
        d_num = None
        self._side = None
        side_ok = self.overlap_ctc and isinstance(model, modules.CTC_Transformer) and not isinstance(model, modules.CIF_Model)
        if side_ok:
            # the decoder's target bookkeeping has one host sync (max target length): do it now, before the step is queued
            ev = None
            if max_target_len is not None and targets.is_cuda:
                # ... and with the length known there is no sync at all: the ~15 small index kernels go to a side stream beside
                # the encoder forward (first needed at the decoder / the CTC branch, which wait for the event)
                main, aux = torch.cuda.current_stream(), ops.aux_stream(targets.device, slot=4)
                aux.wait_stream(main)
                with torch.cuda.stream(aux):
                    pre = model.decoder._preprocess(targets, umax=max_target_len)
                for t in pre:
                    t.record_stream(main)
                ev = torch.cuda.Event()
                ev.record(aux)
            else:
                pre = model.decoder._preprocess(targets, umax=max_target_len)
            model.decoder.__dict__["_pre_hint"] = (targets, pre, ev)
            model.__dict__["_ctc_hook"] = lambda enc, l: self._ctc_side_branch(enc, l, pre[1], ev)
        # every other family with the reference's attention decoder (the attention-only Transformer; the joint models with the CTC branch
        # inline): the loader's longest target goes to the decoder's target bookkeeping - no host read-back in the step, so it captures
        dec_hint = (not side_ok and max_target_len is not None and targets.is_cuda and
                    isinstance(getattr(model, "decoder", None), modules.Decoder) and not isinstance(model, modules.CIF_Model))
        if dec_hint:
            model.decoder.__dict__["_pre_hint"] = (targets, model.decoder._preprocess(targets, umax=max_target_len), None)
        cif_hint = isinstance(model, modules.CIF_Model) and max_target_len is not None
        if cif_hint:
            model.__dict__["_umax_hint"] = int(max_target_len)
        try:
            return self._forward_loss(feats, lens, targets, noise)
        finally:
            if cif_hint:
                model.__dict__.pop("_umax_hint", None)
            if side_ok:
                model.__dict__.pop("_ctc_hook", None)
                model.decoder.__dict__.pop("_pre_hint", None)
            if dec_hint:
                model.decoder.__dict__.pop("_pre_hint", None)

    def _forward_loss(self, feats, lens, targets, noise):
        model = self.model
        d_num = None
        with torch.no_grad(), modules.record() as tape:
            if isinstance(model, CTC_Model):
                # pure-CTC family (ctcModel/solver.py:30): loss = cal_loss(logits, len_logits, targets) - CTC on the raw targets
                logits, ctc_len = model(feats, lens)
                ctc, nll, st = ops.ctc_loss_fwd(logits, ops.as_i32(ctc_len, logits.device), targets)
                return ctc, torch.zeros((), device=logits.device), (tape, st, None, None, None, None, None)
            if isinstance(model, modules.CIF_Model):
                # CIF family (solver.py:146-153): CTC and CE both on `targets` (no <eos>), plus the quantity loss on sum(alpha)
                ctc_logits, ctc_len, num_pred, num, logits = model(feats, lens, targets, noise=noise)
                teos = targets
                d_num = self.lambda_qua * 2.0 * (num_pred - num) / num.numel()
                self.last_qua = ((num_pred - num) ** 2).mean()
            else:
                out = model(feats, lens, targets)
                if isinstance(model, modules.Conv_CTC_Transformer):
                    ctc_logits, ctc_len, logits, teos = out
                elif isinstance(model, modules.CTC_Transformer):
                    ctc_len, ctc_logits, (logits, teos) = out
                else:
                    # attention-only family (Transformer_Solver, solver.py:26-31): loss = cal_ce_loss(logits, targets_eos) alone
                    (logits, teos), ctc_logits = out, None
            if ctc_logits is None:
                ctc, st = torch.zeros((), device=logits.device), None
            elif self._side is not None:     # already computed (and differentiated) on the side stream
                ctc, st = self._side["ctc"], None
            else:
                ctc, nll, st = ops.ctc_loss_fwd(ctc_logits, ops.as_i32(ctc_len, ctc_logits.device), teos)
            V = logits.shape[-1]
            loss2, row_loss, lse, tg1 = ops.ce_loss_fwd(logits.reshape(-1, V), teos.reshape(-1), self.smoothing)
            self._nw_handle = None
            if (self.world > 1 or self.buckets.mark_stream is not None) and self.exact_global_mean:
                nw = loss2[1:2].clone()
                if self.buckets.mark_stream is not None:      # captured for the executor: a collective node on the comm stream
                    ms = self.buckets.mark_stream
                    ms.wait_stream(torch.cuda.current_stream())
                    with torch.cuda.stream(ms):
                        ops.collective_mark(nw, tag=-1)
                    nw.record_stream(ms)
                    self._nw_handle, self._nw_marked = (None, nw), nw
                else:
                    self._nw_handle = (torch.distributed.all_reduce(nw, group=self.group, async_op=True), nw)
        return ctc, loss2[0], (tape, st, logits, tg1, lse, loss2, d_num)

    def backward(self, state):
        """Replays the tape.  The layers' weight-gradient GEMMs (dW = dY^T . X: MFMA work plus a float-atomic epilogue, nothing
        downstream in the backward reads them) are queued on a side stream (`modules._wg`) and co-run with the HBM / VALU-bound
        kernels of the main chain - LayerNorm backward, attention backward, mask hashing; a finished bucket's all-reduce is
        launched from that stream, so it is ordered after the weight gradients it sums.  15.24 -> 14.39 ms per step at S1.
        (The same idea cost +2 ms in the first version of this trainer, when every GEMM was a one-tile-per-workgroup launch with
        a run-time epilogue; giving the main chain a high-priority stream on top of it loses the whole gain again.)"""
        tape, st, logits, tg1, lse, loss2, d_num = state
        dev = self.fp.flat.device
        wg = self.wgrad_stream and dev.type == "cuda"
        if wg:
            side = ops.aux_stream(dev, slot=2)
            modules._WGRAD = {"stream": side, "keep": []}
            self.buckets.launch_stream = side
        try:
            self._backward(tape, st, logits, tg1, lse, loss2, d_num)
            modules.flush_wgrads()
        finally:
            if wg:
                torch.cuda.current_stream().wait_stream(side)      # before Adam reads the gradients (and before `keep` is dropped)
                modules._WGRAD = None
                self.buckets.launch_stream = None

    def _backward(self, tape, st, logits, tg1, lse, loss2, d_num):
        model = self.model
        one = self.__dict__.get("_one")      # (a constant: not a fill launch per step on the decoder's chain)
        if one is None or one.device != self.fp.flat.device:
            if torch.cuda.is_available() and torch.cuda.is_current_stream_capturing():
                one = torch.ones(1, device=self.fp.flat.device)
            else:
                one = self._one = torch.ones(1, device=self.fp.flat.device)
        with torch.no_grad():
            if logits is None:     # CTC_Model: the CTC loss is the whole objective
                model.decoder._grad_slots["prj"]["g"] = ops.ctc_loss_bwd(st, one, bf16=(modules.get_precision() == "bf16"))
                self.buckets.start()
                tape.backward(lambda fn: self.buckets.on_done(fn.params))
                self.buckets.finish()
                return
            # loss = ctc + ce (solver.py:88): both seeds are 1
            if st is not None:
                model._grad_slots["ctc"]["g"] = ops.ctc_loss_bwd(st, one, bf16=(modules.get_precision() == "bf16"))
            V = logits.shape[-1]
            ce_seed = one
            if self._nw_handle is not None:    # gradient of the global-batch CE mean: seed = world * n_word_local / n_word_global
                h, nw = self._nw_handle
                if h is None:
                    torch.cuda.current_stream().wait_stream(self.buckets.mark_stream)
                else:
                    h.wait()
                ce_seed = loss2[1:2] * (float(self.world) / nw)
                self._nw_handle = None
            model.decoder._grad_slots["prj"]["g"] = ops.ce_loss_bwd(logits.reshape(-1, V), tg1, self.smoothing, lse, loss2, ce_seed, bf16=(modules.get_precision() == "bf16"))
            if d_num is not None:
                model._grad_slots["num"]["g"] = d_num
            self.buckets.start()
            tape.backward(lambda fn: self.buckets.on_done(fn.params))
            self.buckets.finish()

    def optimizer_step(self):
        self.step_num += 1
        ops.adam_step(self.fp.flat, self.fp.grad, self.m, self.v, self.lr(), self.betas[0], self.betas[1], self.eps, self.step_num,
                      grad_scale=1.0 / self.world, p16=self.fp.flat16)
        modules.bump_param_epoch()     # derived (re-laid-out) weights are rebuilt from the new parameters on next use

    def _fwd_bwd(self, feats, lens, targets, noise, max_target_len):
        self.fp.grad.zero_()
        if feats.is_cuda:
            ops.arena_reset(feats.device)      # pre-zeroed outputs for the step's split-K GEMMs (ops: zero arena)
        try:
            ctc, ce, state = self.forward_loss(feats, lens, targets, noise=noise, max_target_len=max_target_len)
            self.backward(state)
        except BaseException:
            ops.ctc_reset_counters()      # an aborted step may leave the one-launch CTC forward's arrival counters non-zero
            raise
        finally:
            ops.arena_release()
        return ctc, ce

    def losses_finite(self, ctc, ce):
        """Host read of a step's losses (a sync - call it where the losses are logged, solver.py:52-56): False for a non-finite value,
        after forgetting the CTC forward's cached arrival counters (a bounded wait that gave up returns NaN and leaves them stale)."""
        ok = bool(torch.isfinite(ctc).all()) and bool(torch.isfinite(ce).all())
        if not ok:
            ops.ctc_reset_counters()
        return ok

    def step(self, feats, lens, targets, noise=None, max_target_len=None):
        """One full training step; returns (ctc_loss, ce_loss) tensors (no host sync when max_target_len is given)."""
        ctc, ce = self._fwd_bwd(feats, lens, targets, noise, max_target_len)
        self.optimizer_step()
        return ctc, ce

    # ---- the fixed-shape step as ONE hipGraph ----------------------------------------------------------------------------------
    # A step is ~550 kernel launches on four streams; queued eagerly the host spends ~9 ms per step in ctypes / torch calls.  With
    # fixed shapes (a bucketed loader, the synthetic bench) the whole step - zeroing, forward, losses, backward on its side
    # streams, Adam - is captured once and replayed with one call.  What changes from step to step lives in device memory the
    # graph reads: the step state of asr_step_tick (step number -> Noam lr, Adam bias corrections) and, through
    # asr_dropout_t.salt, the dropout masks (a fresh mask per step from the same descriptors).
    def graph_active(self):
        return self._graph is not None

    def _graph_ok(self, feats, max_target_len):
        if self._graph_failed is not None or not feats.is_cuda or max_target_len is None:
            return False
        if self.world > 1 and (not GRAPH_EXEC or os.environ.get("ASR_AMD_GRAPH_DP", "0") != "1"):
            # with N > 1 the step is captured only for the executor (its C loop makes the RCCL calls), and only on request: the
            # library-owned communicator has run at world = 1 and through the gloo rig, never on N > 1 GPUs - until it has,
            # N > 1 steps are queued eagerly and all-reduced through torch.distributed
            return False
        m = self.model
        return isinstance(m, (modules.Transformer, modules.CIF_Model)) and modules.get_precision() == "bf16"      # (every tape-recording family)

    def _sync_state(self, dev):
        """device step state <- host step counter (first use, or after eager steps / a checkpoint load moved it)"""
        if self._state is None:
            self._state = torch.zeros(8, dtype=torch.int32, device=dev)
        if self._state_step != self.step_num:
            self._state.copy_(torch.tensor([self.step_num] + [0] * 7, dtype=torch.int32))
            self._state_step = self.step_num

    def step_graphed(self, feats, lens, targets, noise=None, max_target_len=None):
        """`step` for a loader with fixed shapes: the first two calls run eagerly (allocator pools, code objects, side streams), the
        third captures the step into a hipGraph, later calls with inputs of the same shapes replay it (the batch is copied into the
        buffers the graph was captured against).  Falls back to `step` (and says why in `self._graph_failed`) when the step cannot be
        captured.  The returned (ctc, ce) tensors are the graph's own outputs: the next replay overwrites them."""
        if not self._graph_ok(feats, max_target_len):
            return self.step(feats, lens, targets, noise=noise, max_target_len=max_target_len)
        if noise is None and getattr(self.model, "draws_noise", False):
            # CIF_Model's per-utterance noise (cif_model.py:47) is drawn HERE, outside the graph, and fed in like the batch: a torch.rand
            # inside the captured step replays with the capture-time generator state under the executor (torch's own replay advances
            # the philox offset in its prologue, asr_graphx_launch does not) and every step would see the same vector
            noise = torch.rand(feats.size(0), device=feats.device)
        # The graph holds raw pointers, so it is captured against buffers the trainer owns and every call copies its batch into them
        # (a loader hands out NEW tensors of the same shape each step: keyed on their addresses every call missed the key and
        # re-captured the whole step).  The key is what fixes the graph's shape: shapes / dtypes of every input, the longest target.
        sig = lambda t: None if t is None else (tuple(t.shape), t.dtype)
        key = (sig(feats), sig(lens), sig(targets), sig(noise), max_target_len, self.model.training, self.overlap_ctc, self.wgrad_stream)
        if self._graph is not None and self._graph_key != key and self.world > 1 and self._graphx is not None:
            # A re-capture is a collective act (the communicator bootstrap and the MIN "ok" all-reduce of _capture run through
            # torch.distributed) while the peers that still hold a matching graph sit in the executor's all-reduces on the library's own
            # communicator: the two sides would wait on different communicators forever.  Fixed shapes on every rank are the contract of
            # the captured data-parallel step; anything else has to say so loudly.
            raise RuntimeError("Trainer.step_graphed with world_size %d: the batch signature changed after the step was captured (%r -> %r). "
                               "The captured data-parallel step needs the same shapes and max_target_len on every rank and every step - "
                               "bucket the loader to fixed shapes, or step eagerly (Trainer.step; unset ASR_AMD_GRAPH_DP - eager is the default for world_size > 1)."
                               % (self.world, self._graph_key, key))
        if self._graph is None or self._graph_key != key:
            if self._eager_steps < 2:
                self._eager_steps += 1
                return self.step(feats, lens, targets, noise=noise, max_target_len=max_target_len)
            self._graph_in = tuple(None if t is None else t.detach().clone() for t in (feats, lens, targets, noise))
            self._capture(*self._graph_in, max_target_len, key)
            if self._graph is None:
                self._graph_in = None
                return self.step(feats, lens, targets, noise=noise, max_target_len=max_target_len)
            # (the capture pass above does not execute: the replay below is this call's step)
        for dst, src in zip(self._graph_in, (feats, lens, targets, noise)):
            if dst is not None and dst.data_ptr() != src.data_ptr():
                dst.copy_(src, non_blocking=True)
        self._sync_state(feats.device)
        if self._graphx is not None:
            self._graphx.launch()
        else:
            self._graph.replay()
        self.step_num += 1
        self._state_step = self.step_num
        modules.bump_param_epoch()
        return self._graph_out

    def step_auto(self, feats, lens, targets, noise=None, max_target_len=None, trials=4):
        """`step` that picks its own launch mode for a fixed-shape loader.  Which one wins depends on the workload: with long
        utterances (S1, L = 1000) the step is GPU-bound and the eager launches on four free-running streams overlap better than
        the replayed graph's branches (13.5 vs 15.1 ms); with the conv front end (S2, L = 250) the kernels are 10-30 us, the host
        cannot queue them fast enough and the replay wins (8.1 vs 12.4 ms).  The first calls are the calibration - `trials` eager
        steps, capture, `trials` replays, all of them real training steps - and every later call uses the faster mode
        (`self.launch_mode`, timings in `self.launch_timing`)."""
        if self.launch_mode is None:
            if not self._graph_ok(feats, max_target_len):
                self.launch_mode = "eager"
            else:
                return self._calibrate(feats, lens, targets, noise, max_target_len, trials)
        if self.launch_mode == "graph":
            return self.step_graphed(feats, lens, targets, noise=noise, max_target_len=max_target_len)
        return self.step(feats, lens, targets, noise=noise, max_target_len=max_target_len)

    def _calibrate(self, feats, lens, targets, noise, max_target_len, trials):
        import time
        dev = feats.device

        def timed(fn):
            torch.cuda.synchronize(dev)
            t0 = time.perf_counter()
            for _ in range(trials):
                out = fn(feats, lens, targets, noise=noise, max_target_len=max_target_len)
            torch.cuda.synchronize(dev)
            ms = (time.perf_counter() - t0) / trials * 1e3
            if self.world > 1:      # one decision for all ranks (a rank replaying while another queues eagerly would wait on different communicators)
                t = torch.tensor([ms], device=dev)
                torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX, group=self.group)
                ms = float(t)
            return ms, out

        def host_us(fn):
            """the host's own queueing time of one step: the call's duration with the device idle at its start (with work queued
            ahead the call would mostly measure the runtime's back-pressure)"""
            tot = 0.0
            for _ in range(3):
                torch.cuda.synchronize(dev)
                t0 = time.perf_counter()
                fn(feats, lens, targets, noise=noise, max_target_len=max_target_len)
                tot += time.perf_counter() - t0
            torch.cuda.synchronize(dev)
            return round(tot / 3 * 1e6, 1)
        for _ in range(2):                               # pools, code objects, side streams
            self.step(feats, lens, targets, noise=noise, max_target_len=max_target_len)
        t_eager, out = timed(self.step)
        h_eager = host_us(self.step)
        self._eager_steps = 2
        out = self.step_graphed(feats, lens, targets, noise=noise, max_target_len=max_target_len)      # captures and replays once
        if not self.graph_active():
            self.launch_mode, self.launch_timing = "eager", {"eager_ms": round(t_eager, 3), "graph_ms": None, "eager_host_us": h_eager}
            return out
        t_graph, out = timed(self.step_graphed)
        rot = None
        if self._graphx is not None and self._graphx.info["streams"] > 2:
            # the executor's side streams share hardware queues with each other and with the launch stream; which ones do is the
            # runtime's choice (creation order of every stream in the process) and decides whether the weight-gradient chain overlaps
            # the main chain at all: time the rotations of the logical -> physical stream map, keep the best (real steps all of them)
            rot = {0: round(t_graph, 3)}
            best = 0
            for r in range(1, min(4, self._graphx.info["streams"] - 1)):
                self._graphx.set_rotation(r)
                t_r, out = timed(self.step_graphed)
                rot[r] = round(t_r, 3)
                if t_r < t_graph:
                    t_graph, best = t_r, r
            self._graphx.set_rotation(best)
            if ops.QUEUE_PROBE:          # ... and the placement computed from hardware-queue probes against the best rotation
                torch.cuda.synchronize(dev)
                self._graphx.place_streams()
                t_p, out = timed(self.step_graphed)
                rot["placed"] = round(t_p, 3)
                if t_p < t_graph:
                    t_graph = t_p
                else:
                    self._graphx.place_streams(clear=True)
        dec = getattr(self.model, "decoder", None)
        if dec is not None and hasattr(dec, "target_overflow") and dec.target_overflow():
            raise RuntimeError("asr_amd.Trainer: max_target_len=%r is smaller than the longest target of the batch (a target was truncated)" % (max_target_len,))
        self.launch_mode = "graph" if t_graph < t_eager else "eager"
        self.launch_timing = {"eager_ms": round(t_eager, 3), "graph_ms": round(t_graph, 3), "eager_host_us": h_eager,
                              "graph_host_us": host_us(self.step_graphed)}
        if self._graphx is not None and self._graphx.info.get("collectives"):
            self.launch_timing["collective_nodes"] = self._graphx.info["collectives"]
        if rot is not None:
            self.launch_timing["graph_ms_by_stream_rotation"] = rot
        if self.launch_mode == "eager":
            self._graphx = None
            self._graph = None                           # frees the graph's private pool
        return out

    def _capture(self, feats, lens, targets, noise, max_target_len, key):
        dev = feats.device
        self._graphx = None
        self._graph = None
        self._sync_state(dev)
        modules.bump_param_epoch()             # derived weights are rebuilt inside the capture (and so at every replay)
        torch.cuda.synchronize(dev)
        # keep_graph: the captured hipGraph stays readable - its nodes are what the multi-stream executor launches (ops.GraphExec);
        # a stack without that option (or a graph the executor does not take) replays through hipGraphLaunch
        use_x = GRAPH_EXEC
        # With more than one rank (or force_collective) the gradient all-reduces are NODES of the captured step: GradBuckets marks each
        # bucket's ready point on a comm stream of its own, the executor's launch loop calls RCCL there (csrc/collective.hip).  Such a
        # step only replays through the executor - under hipGraphLaunch the markers would be no-ops and the ranks would drift apart.
        collective = self.world > 1 or self.force_collective
        try:
            g = torch.cuda.CUDAGraph(keep_graph=True) if use_x else torch.cuda.CUDAGraph()
        except TypeError:
            g, use_x = torch.cuda.CUDAGraph(), False
        self._graphx = None
        modules._DROP_STATE["salt"] = self._state.data_ptr()
        try:
            comm, comm_fn = None, None
            if collective:
                if not use_x:
                    raise RuntimeError("a data-parallel step is captured for the graph executor only (trainer.GRAPH_EXEC off or no keep_graph)")
                # (communicator set-up is itself a collective and queues GPU work: before the capture begins)
                if self.world > 1 and torch.distributed.get_backend(self.group) != "nccl" and os.environ.get("ASR_AMD_DP_COMM", "") != "rccl":
                    comm_fn = ops.torch_collective_fn([self.fp.grad], self.group)      # the rig: gloo ranks sharing one GPU
                else:
                    comm = ops.rccl_comm(self.group, dev)
                self.buckets.mark_stream = ops.aux_stream(dev, slot=3)
            # (N > 1: torch's process group has a watchdog thread that queries events while this thread captures; its calls must not
            # invalidate the capture)
            cap_kw = {"capture_error_mode": "thread_local"} if self.world > 1 else {}
            with torch.cuda.graph(g, **cap_kw):
                ops.step_tick(self._state, self.k, self.init_lr, self.warmup, self.betas[0], self.betas[1])
                ctc, ce = self._fwd_bwd(feats, lens, targets, noise, max_target_len)
                ops.adam_step_dev(self.fp.flat, self.fp.grad, self.m, self.v, self._state, self.betas[0], self.betas[1], self.eps,
                                  grad_scale=1.0 / self.world, p16=self.fp.flat16)
            if use_x:
                self._graphx = ops.GraphExec.from_torch_graph(g)       # None (with a warning) when the executor does not take this graph
            if collective:
                if self._graphx is None:
                    raise RuntimeError("the graph executor did not take the data-parallel step")
                if self._graphx.info["collectives"] < len(self.buckets.ranges):
                    raise RuntimeError("the executor found %d collective nodes for %d gradient buckets" %
                                       (self._graphx.info["collectives"], len(self.buckets.ranges)))
                if comm_fn is not None and self._nw_marked is not None:
                    comm_fn = ops.torch_collective_fn([self.fp.grad, self._nw_marked], self.group)
                self._graphx.set_collective(comm=comm, fn=comm_fn)
            self._graph, self._graph_key, self._graph_out = g, key, (ctc, ce)
        except Exception as e:      # not capturable on this stack: keep training eagerly, remember why
            self._graph_failed = "%s: %s" % (type(e).__name__, e)
            self._graphx = None
            import warnings
            warnings.warn("asr_amd.Trainer: hipGraph capture of the step failed, running eagerly (%s)" % self._graph_failed)
            try:
                torch.cuda.synchronize(dev)
            except Exception:
                pass
        finally:
            modules._DROP_STATE["salt"] = None
            self.buckets.mark_stream = None
        if self.world > 1:
            # every rank replays, or none does: a rank that fell back to the eager step would all-reduce through torch's communicator
            # while the others wait in the executor's
            ok = torch.tensor([1.0 if self._graph is not None else 0.0], device=dev)
            torch.distributed.all_reduce(ok, op=torch.distributed.ReduceOp.MIN, group=self.group)
            if float(ok) == 0.0 and self._graph is not None:
                self._graph, self._graphx = None, None
                self._graph_failed = "another rank could not capture the step"
