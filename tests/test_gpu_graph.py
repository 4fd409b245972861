"""The training step replayed from a captured hipGraph (Trainer.step_graphed) against the same step queued eagerly: same losses,
same parameter trajectory (to the float-atomic summation noise of the weight-gradient kernels), the device step state follows
the Noam schedule of src/transformer/optimizer.py:24-29, and dropout draws a fresh mask on every replay."""
import os

import numpy as np
import pytest
import torch

import asr_amd
from oracle import asr_oracle as O
from weights import make_state_dict, names_shapes_from_json

pytestmark = pytest.mark.gpu
DEV = "cuda:0"


def build(golden_dir, dropout=0.0):
    z = np.load(os.path.join(golden_dir, "g1_ctc_transformer.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=dropout),
                                    asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=dropout))
    model.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    return z, model.to(DEV).train()


def test_graph_replay_matches_eager_steps(golden_dir):
    asr_amd.set_precision("bf16")
    z, m_e = build(golden_dir)
    _, m_g = build(golden_dir)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    # (a gentle schedule: the weight-gradient kernels sum with float atomics, so two runs differ in the last bits, and with
    # warmup 5 the loss surface amplifies that to 0.7 % within three steps)
    te = asr_amd.Trainer(m_e, k=0.2, warmup_steps=50, label_smoothing=0.1)
    tg_ = asr_amd.Trainer(m_g, k=0.2, warmup_steps=50, label_smoothing=0.1)
    le, lg = [], []
    for i in range(8):
        c, e = te.step(x, lens, tg, max_target_len=umax)
        le.append((float(c), float(e)))
        c, e = tg_.step_graphed(x, lens, tg, max_target_len=umax)
        lg.append((float(c), float(e)))
    assert tg_.graph_active(), tg_._graph_failed
    assert tg_.step_num == te.step_num == 8
    np.testing.assert_allclose(np.array(lg), np.array(le), rtol=5e-3)
    pe = te.fp.flat.float().cpu().numpy()
    pg = tg_.fp.flat.float().cpu().numpy()
    assert np.linalg.norm(pg - pe) / np.linalg.norm(pe) < 2e-3
    # device step state: {step, lr, 1 - b1^step, sqrt(1 - b2^step)}
    st = tg_._state.cpu().numpy()
    assert int(st[0]) == 8
    f = st.view(np.float32)
    np.testing.assert_allclose(f[1], O.noam_lr(8, 0.2, 64, 50), rtol=1e-6)
    np.testing.assert_allclose(f[1], te.lr(), rtol=1e-6)
    np.testing.assert_allclose(f[2], 1.0 - 0.9 ** 8, rtol=1e-6)
    np.testing.assert_allclose(f[3], np.sqrt(1.0 - 0.98 ** 8), rtol=1e-6)
    # eager steps after replays (and replays after them) stay on one trajectory: the host counter and the device state re-sync
    te.step(x, lens, tg, max_target_len=umax)
    tg_.step(x, lens, tg, max_target_len=umax)
    c1, e1 = te.step(x, lens, tg, max_target_len=umax)
    c2, e2 = tg_.step_graphed(x, lens, tg, max_target_len=umax)
    assert tg_.step_num == te.step_num == 10 and int(tg_._state.cpu()[0]) == 10
    np.testing.assert_allclose([float(c2), float(e2)], [float(c1), float(e1)], rtol=5e-3)


def test_graph_replay_draws_new_dropout_masks(golden_dir):
    asr_amd.set_precision("bf16")
    z, model = build(golden_dir, dropout=0.3)
    asr_amd.manual_seed(11)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    tr = asr_amd.Trainer(model, k=0.0, warmup_steps=5, label_smoothing=0.1)     # k = 0: lr 0, the weights never move
    losses = []
    for i in range(7):
        c, e = tr.step_graphed(x, lens, tg, max_target_len=umax)
        losses.append(float(c) + float(e))
    assert tr.graph_active(), tr._graph_failed
    replayed = losses[3:]          # calls 0, 1 eager, call 2 captures + replays
    assert len(set(round(v, 5) for v in replayed)) == len(replayed), losses      # same weights, different masks -> different losses
    assert np.std(replayed) < 0.2 * np.mean(replayed)


def test_dropout_salt_resolves_like_the_header_says():
    from asr_amd import ops
    x = torch.randn(3, 5, 64, device=DEV)
    salt = torch.tensor([12345], dtype=torch.int32, device=DEV)
    thr, k0, k1 = 19661, 0x1234567, 0x89abcde
    y = ops.dropout_apply(x.clone(), ops.Dropout(thr, k0, k1, salt.data_ptr()), 3, 5, 64).cpu().numpy()

    def lowbias32(v):
        v &= 0xFFFFFFFF
        v ^= v >> 16
        v = (v * 0x7feb352d) & 0xFFFFFFFF
        v ^= v >> 15
        v = (v * 0x846ca68b) & 0xFFFFFFFF
        v ^= v >> 16
        return v
    r0 = k0 ^ lowbias32(12345 ^ 0x5bd1e995)
    r1 = k1 ^ lowbias32((12345 + 0x27d4eb2f) & 0xFFFFFFFF)
    exp = x.cpu().numpy() * O.dropout_mask((3, 5, 64), thr, r0, r1)      # (the mask carries the 1/keep scale)
    np.testing.assert_allclose(y, exp, rtol=1e-6, atol=0)


def test_step_auto_calibrates_once_and_stays_on_the_eager_trajectory(golden_dir):
    asr_amd.set_precision("bf16")
    z, m_e = build(golden_dir)
    _, m_a = build(golden_dir)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    te = asr_amd.Trainer(m_e, k=0.2, warmup_steps=50, label_smoothing=0.1)
    ta = asr_amd.Trainer(m_a, k=0.2, warmup_steps=50, label_smoothing=0.1)
    ta.step_auto(x, lens, tg, max_target_len=umax, trials=2)          # calibration: 2 + 2 eager, 1 capture + replay, 2 replays, and 2 more
    n_cal = ta.step_num                                               # per stream rotation / placement the executor is timed with;
    assert ta.launch_mode in ("eager", "graph") and n_cal >= 13 and (n_cal - 13) % 2 == 0      # 3 + 3 for the host's queueing time
    assert ta.launch_timing["eager_ms"] > 0 and ta.launch_timing["graph_ms"] > 0
    assert 0 < ta.launch_timing["graph_host_us"] < ta.launch_timing["eager_host_us"]          # (the C loop against ~600 Python calls)
    assert ta.graph_active() == (ta.launch_mode == "graph")
    for _ in range(n_cal):
        ce = te.step(x, lens, tg, max_target_len=umax)
    for _ in range(3):
        ce = te.step(x, lens, tg, max_target_len=umax)
        ca = ta.step_auto(x, lens, tg, max_target_len=umax)
    assert ta.step_num == te.step_num == n_cal + 3
    np.testing.assert_allclose([float(v) for v in ca], [float(v) for v in ce], rtol=5e-3)
    # a step that cannot be captured (no max_target_len: a host read in the middle) settles on eager without calibrating
    tb = asr_amd.Trainer(build(golden_dir)[1], k=0.2, warmup_steps=50, label_smoothing=0.1)
    tb.step_auto(x, lens, tg)
    assert tb.launch_mode == "eager" and tb.step_num == 1


def test_cif_model_step_replays_from_a_graph(golden_dir):
    """CIF family: with the longest target given the step has no host read-back (max_label_len = the longest target, cif_model.py:44-48
    make round(sum alpha) = num) and is capturable; replayed steps follow the eager trajectory (same recorded noise)."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    asr_amd.set_precision("bf16")

    def fresh():
        m = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
        m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
        return m.to(DEV).eval()
    x, lens, tg, noise = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets", "noise"))
    umax = int((tg != 0).sum(1).max())
    te = asr_amd.Trainer(fresh(), k=0.2, warmup_steps=50, label_smoothing=0.1, lambda_qua=0.001)
    tg_ = asr_amd.Trainer(fresh(), k=0.2, warmup_steps=50, label_smoothing=0.1, lambda_qua=0.001)
    le, lg = [], []
    for i in range(6):
        le.append([float(v) for v in te.step(x, lens, tg, noise=noise)])                       # (reads max_label_len back, like the reference)
        lg.append([float(v) for v in tg_.step_graphed(x, lens, tg, noise=noise, max_target_len=umax)])
    assert tg_.graph_active(), tg_._graph_failed
    np.testing.assert_allclose(np.array(lg), np.array(le), rtol=5e-3)
    pe, pg = te.fp.flat.float().cpu().numpy(), tg_.fp.flat.float().cpu().numpy()
    assert np.linalg.norm(pg - pe) / np.linalg.norm(pe) < 2e-3


def test_graph_replay_takes_fresh_input_tensors(golden_dir):
    """A loader hands out new tensors every step: the captured graph is keyed on shapes, the batch is copied into the buffers it
    was captured against - no re-capture, and the replay sees the NEW data (CIF's noise tensor included in the key)."""
    asr_amd.set_precision("bf16")
    z, m_e = build(golden_dir)
    _, m_g = build(golden_dir)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    te = asr_amd.Trainer(m_e, k=0.2, warmup_steps=50, label_smoothing=0.1)
    tg_ = asr_amd.Trainer(m_g, k=0.2, warmup_steps=50, label_smoothing=0.1)
    g = torch.Generator().manual_seed(5)
    captures = []
    orig = tg_._capture
    tg_._capture = lambda *a, **k: (captures.append(1), orig(*a, **k))[1]
    for i in range(7):
        xi = (x + 0.3 * torch.randn(x.shape, generator=g).to(DEV)).contiguous()      # a different batch in a different allocation
        li, ti = lens.clone(), tg.clone()
        ce, ee = te.step(xi, li, ti, max_target_len=umax)
        cg, eg = tg_.step_graphed(xi, li, ti, max_target_len=umax)
        np.testing.assert_allclose([float(cg), float(eg)], [float(ce), float(ee)], rtol=5e-3)
    assert tg_.graph_active() and len(captures) == 1
    pe, pg = te.fp.flat.float().cpu().numpy(), tg_.fp.flat.float().cpu().numpy()
    assert np.linalg.norm(pg - pe) / np.linalg.norm(pe) < 2e-3


def test_cif_model_replays_draw_fresh_noise(golden_dir):
    """the executor-launched graph skips torch's replay prologue, so a torch.rand INSIDE the captured step would repeat the
    capture-time vector.  step_graphed draws CIF_Model's noise (cif_model.py:47) outside the graph and copies it in: replayed steps
    with noise=None see a fresh vector each."""
    import argparse
    z = np.load(os.path.join(golden_dir, "g4_cif_model.npz"))
    sd = make_state_dict(names_shapes_from_json(z["names_shapes"]), int(z["seed"]))
    cfg = {k[4:]: z[k].item() for k in z.files if k.startswith("cfg_")}
    asr_amd.set_precision("bf16")
    m = asr_amd.CIF_Model.create_model(argparse.Namespace(spec_aug_cfg=None, **cfg))
    m.load_state_dict({k: torch.from_numpy(v) for k, v in sd.items()}, strict=False)
    m = m.to(DEV).train()
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    tr = asr_amd.Trainer(m, k=0.2, warmup_steps=50, label_smoothing=0.1, lambda_qua=0.001)
    seen = []
    for _ in range(6):
        tr.step_graphed(x, lens, tg, max_target_len=umax)
        if tr.graph_active():
            torch.cuda.synchronize()
            seen.append(tr._graph_in[3].clone())
    assert tr.graph_active() and len(seen) >= 3, tr._graph_failed
    assert all(not torch.equal(seen[i], seen[i + 1]) for i in range(len(seen) - 1))
    assert all(float(v.min()) >= 0.0 and float(v.max()) < 1.0 for v in seen)


def test_executor_runs_the_gradient_all_reduce_itself(golden_dir):
    """The data-parallel step's collectives as nodes of the captured step (csrc/collective.hip): with force_collective a 1-rank RCCL
    communicator stands in for the ranks, so every gradient bucket (and the CE word count) is all-reduced by the executor's own C
    loop - same trajectory as the step without them (the sum over one rank is the identity)."""
    asr_amd.set_precision("bf16")
    z, m_a = build(golden_dir)
    _, m_b = build(golden_dir)
    x, lens, tg = (torch.from_numpy(z[k]).to(DEV) for k in ("x", "lens", "targets"))
    umax = int((tg != 0).sum(1).max())
    ta = asr_amd.Trainer(m_a, k=0.2, warmup_steps=50, label_smoothing=0.1)
    tb = asr_amd.Trainer(m_b, k=0.2, warmup_steps=50, label_smoothing=0.1, force_collective=True)
    la, lb = [], []
    for i in range(8):
        c, e = ta.step_graphed(x, lens, tg, max_target_len=umax)
        la.append((float(c), float(e)))
        c, e = tb.step_graphed(x, lens, tg, max_target_len=umax)
        lb.append((float(c), float(e)))
    assert tb.graph_active() and tb._graphx is not None, tb._graph_failed
    info = tb._graphx.info
    assert info["collectives"] == len(tb.buckets.ranges) + 1, info               # every bucket + the CE word count
    assert info["collective_floats"] == tb.fp.grad.numel() + 1, info              # ... the whole flat gradient, once
    assert ta._graphx.info["collectives"] == 0
    np.testing.assert_allclose(np.array(lb), np.array(la), rtol=5e-3)
    pa, pb = ta.fp.flat.float().cpu().numpy(), tb.fp.flat.float().cpu().numpy()
    assert np.linalg.norm(pb - pa) / np.linalg.norm(pa) < 2e-3
    from asr_amd import ops
    ops.rccl_comm().check()


def test_collective_nodes_without_a_communicator_fail_loudly(golden_dir):
    asr_amd.set_precision("bf16")
    from asr_amd import ops
    buf = torch.ones(1024, device=DEV)
    g = torch.cuda.CUDAGraph(keep_graph=True)
    side = torch.cuda.Stream()
    with torch.cuda.graph(g):
        buf.mul_(2.0)
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            ops.collective_mark(buf[:512], tag=3)
        torch.cuda.current_stream().wait_stream(side)
        buf.add_(1.0)
    gx = ops.GraphExec.from_torch_graph(g)
    assert gx is not None and gx.info["collectives"] == 1 and gx.info["collective_floats"] == 512
    with pytest.raises(RuntimeError, match="no communicator"):
        gx.launch()
    torch.cuda.synchronize()
    # ... with a callback in the node's place: called once, on the node's stream, between its predecessor and its successor
    seen = []
    buf.fill_(1.0)

    def fn(ctx, ptr, count, tag, stream):
        seen.append((ptr, count, tag))
        with torch.cuda.stream(torch.cuda.ExternalStream(stream) if stream else torch.cuda.default_stream()):
            buf[:512].mul_(10.0)        # stands in for the sum over ranks
        return 0
    cb = ops._COLLECTIVE_CB(fn)
    cb.state = {"error": None}
    gx.set_collective(fn=cb)
    gx.launch()
    torch.cuda.synchronize()
    assert seen == [(buf.data_ptr(), 512, 3)]
    assert float(buf[0]) == 21.0 and float(buf[600]) == 3.0      # (1 * 2) * 10 + 1 ; (1 * 2) + 1
