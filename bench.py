#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X (contract: see the task statement / DESIGN.md §Measurement).

Workload (BASELINE.json configs[1], SURVEY.md §8d "S1"): CTC_Transformer (no conv front-end), d_model=256 h=4
(d_k=d_v=64) d_inner=2048 enc12/dec6, V=4234, per-GPU batch B=32 x T=1000 x 80-dim fbank (full-length
utterances), U=50, bf16 MFMA operands / fp32 accumulate.  `--model s2` runs SURVEY §8d "S2" instead
(Conv_CTC_Transformer: 2 conv layers, encoder length 250).  One "step" = one pass of the hot path over one
synthetic batch: forward, joint CTC + label-smoothed-CE loss, backward, gradient all-reduce (RCCL), Adam
(`--mode fwd`: eval-mode forward + loss only).
value = input fbank frames per second summed over all ranks (weak scaling: per-GPU batch fixed).

`python bench.py --gpus N` with N > 1 and no torchrun environment starts N ranks itself (one process per GPU, before this
process touches the GPU) and returns their exit code; under `python -m torch.distributed.run ... bench.py --gpus N` it is one
rank and checks WORLD_SIZE == N.

rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel family, timed live with HIP events on the launch
stream), "cpu_baseline" (stock PyTorch CPU running the reference's op sequence - forward + loss + backward - on this node's
host cores, oracle/torch_cpu_ref.py; N=1 only), "kernels" (per-op live timings), "ctc" (the fused CTC loss op alone).
"""
import argparse
import json
import os
import re
import subprocess
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

PEAK_MFMA_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0            # HBM3E spec peak, same table

CFG = dict(d_input=80, d_model=256, n_head=4, d_inner=2048, n_layers_enc=12, n_layers_dec=6, vocab_size=4234,
           sos_id=2, eos_id=3, B=32, T=1000, U=50, n_conv_layers=0)


def parse_args():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=50)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--model", default="s1", choices=["s1", "s2", "cif", "aishell"],
                    help="s1: CTC_Transformer on raw fbank (BASELINE configs[1]); s2: Conv_CTC_Transformer (configs[2], L = T/4); cif: CIF_Model "
                         "(configs[3] in-model: conv front end, 3-layer assigner, integrate-and-fire, Decoder_CIF; --mode train only)")
    ap.add_argument("--mode", default="train", choices=["train", "fwd", "decode"],
                    help="train: forward+loss+backward+grad all-reduce+Adam in train mode (default); fwd: eval-mode forward+loss only; "
                         "decode: eval-mode encoder + greedy batch_decode (KV-cached, --decode-len steps) + CTC greedy decode")
    ap.add_argument("--decode-len", type=int, default=50, help="max_decode_len of --mode decode (decoder.py:138)")
    ap.add_argument("--beam", type=int, default=0, help="--mode decode: beam size; 0 = greedy batch_decode for s1 / s2 (Decoder.batch_beam_decode, "
                         "decoder.py:166, when > 0) and beam 5 for --model cif (Decoder_CIF.recognize_beam, decoder.py:425)")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="dropout rate of the training step (0.1 = every shipped config of the reference, egs/*/conf); ignored by --mode fwd")
    ap.add_argument("--graph", type=int, default=int(os.environ.get("ASR_AMD_GRAPH", "-1")),
                    help="launch mode of the fixed-shape training step.  -1 (default): Trainer.step_auto times a few eager steps against "
                         "a few replays of the captured HIP graph during initialisation and keeps the faster (S1: eager, the four "
                         "free-running streams overlap better than the graph's branches, 13.5 vs 15.1 ms; S2: graph, the 10-30 us "
                         "kernels of L = 250 are launch-bound, 8.1 vs 12.4 ms); 1: always replay; 0: always eager")
    ap.add_argument("--ragged", action="store_true", help="per-utterance lengths U{T/2..T} (max forced to T) and targets U{U/2..U}")
    ap.add_argument("--brief", action="store_true",
                    help="headline timing only: no per-kernel pass, no CPU baseline, no extra legs (what the `also` children of the default run use)")
    ap.add_argument("--per-op", default="both", choices=["both", "in_step"],
                    help="the per-op timing passes after the timed region: both = one with the step's side streams (in step) and one with "
                         "everything inlined on one stream (alone); in_step = only the first, so that EVERY step of the process ran with its "
                         "side streams - the run tools/profile_r5.sh puts under rocprofv3 for kernel_stats.csv")
    ap.add_argument("--side-budget", type=int, default=-1,
                    help="Trainer.side_budget: launch budget (CUs) of the CTC branch's backward on the side stream; 0 = full grids, -1 = the trainer's default")
    ap.add_argument("--trainer-set", action="append", default=[], metavar="NAME=INT",
                    help="A/B runs: set a Trainer attribute (fused_ce=0, side_budget=96, ...) before the first step")
    ap.add_argument("--no-also", action="store_true",
                    help="skip the extra legs of the default 1-GPU run (S2, CIF_Model, greedy decode: BASELINE configs[2] / [3] and SURVEY 8(f)1)")
    return ap.parse_args()


def spawn_ranks(args):
    """`python bench.py --gpus N` outside torchrun: start the N ranks as children (this process has not touched the GPU) and
    pass their exit code on."""
    import socket
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(args.gpus), "--master-addr", "127.0.0.1",
           "--master-port", str(port), os.path.abspath(__file__)] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get("HSA_ENABLE_IPC_MODE_LEGACY", "0"))
    return subprocess.run(cmd, env=env, cwd=ROOT).returncode


def what_name(args, train):
    if train:
        return "training step (train mode, dropout %g): forward + joint CTC/CE loss + backward + grad all-reduce + Adam" % args.dropout
    if args.mode == "decode" and CFG.get("cif"):
        return ("CIF_Model.batch_recognize: conv + encoder + assigner + CIF (target_num %d) + one batched Decoder_CIF beam search "
                "(beam %d, K/V caches, step replayed as a hipGraph)" % (CFG["U"], args.beam))
    if args.mode == "decode" and args.beam > 0:
        return "eval-mode encoder + batch_beam_decode (beam %d, %d steps, K/V caches, replayed step) + CTC greedy decode" % (args.beam, args.decode_len)
    if args.mode == "decode":
        return "eval-mode encoder + greedy batch_decode (%d steps, KV cache) + CTC greedy decode" % args.decode_len
    return "eval-mode forward + joint CTC/CE loss"


def model_name():
    if CFG.get("aishell"):
        return "AISHELL recipe width: CTC_Transformer on LFR-stacked fbank (LFR_m=4, LFR_n=3: 320-d frames, T=%d)" % CFG["T"]
    if CFG.get("cif"):
        return "S3-in-model: CIF_Model (2 conv layers, L=%d, 3-layer assigner, threshold 0.95, loss = 0.001 qua + ctc + ce)" % (CFG["T"] // 4)
    return ("S2: Conv_CTC_Transformer (2 conv layers, L=%d)" % (CFG["T"] // 4)) if CFG["n_conv_layers"] else "S1: CTC_Transformer"


def workload_name(args, train):
    return "%s d_model=%d h=%d d_inner=%d enc%d/dec%d V=%d, per-GPU B=%d x T=%d x %d fbank%s, U=%d, %s" % (
        model_name(), CFG["d_model"], CFG["n_head"], CFG["d_inner"], CFG["n_layers_enc"], CFG["n_layers_dec"], CFG["vocab_size"], CFG["B"],
        CFG["T"], CFG["d_input"], " (ragged lengths)" if args.ragged else "", CFG["U"], what_name(args, train))


def settle_groups(timed_group, world, dev, max_groups=8, rel=0.02):
    """Run `timed_group()` (one group of steps -> ms per step) until two consecutive groups agree to `rel`, at most `max_groups` times; ->
    the list of group times.  With world > 1 every rank sees the MAX over ranks of each group time, so ALL ranks run the same number of
    groups: every step holds collectives, and a rank that settled a group earlier than its peers left them waiting in an all-reduce
    it never joined while it sat in the next barrier - the "hangs once in a few dozen runs" of the multi-rank rig
    (tools/dp_hang_hunt.py: 6 of 80 runs; Python stacks: one rank in barrier(), the other in step())."""
    import torch
    out = []
    for _ in range(max_groups):
        ms = timed_group()
        if world > 1:
            t = torch.tensor([ms], device=dev, dtype=torch.float64)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            ms = float(t)
        out.append(ms)
        if len(out) >= 2 and abs(out[-1] - out[-2]) <= rel * out[-2]:
            break
    return out


def launch_name(args, graphed, trainer=None):
    if args.mode == "decode":
        return "per-token step replayed as a hip-graph, encoder eager" if os.environ.get("ASR_AMD_DECODE_GRAPH", "1") != "0" else "eager"
    if graphed and trainer is not None and getattr(trainer, "_graphx", None) is not None:
        info = trainer._graphx.info
        tail = ", %d gradient all-reduce nodes called from its launch loop" % info["collectives"] if info.get("collectives") else ""
        return "captured step launched by the multi-stream graph executor (%d nodes on %d streams, %d events%s)" % (
            info["nodes"], info["streams"], info["events"], tail)
    return "hip-graph replay" if graphed else "eager"


ALSO_LEGS = (("s2", ["--model", "s2", "--mode", "train"]), ("cif", ["--model", "cif", "--mode", "train"]),
             ("decode_s1", ["--model", "s1", "--mode", "decode"]), ("aishell", ["--model", "aishell", "--mode", "train"]))


def run_also_legs(args):
    """The other workloads BASELINE.json names (configs[2] Conv_CTC_Transformer, configs[3] CIF_Model in-model) and greedy decoding, one
    after the other as CHILD processes of this one - started, and finished, before this process makes its first GPU call (a process
    that has initialised the GPU must not be replaced, and two benchmarks must not share the chip).  Each child is `bench.py --brief`
    with the same steps / warmup; its JSON line is cut down to the numbers that matter here."""
    out = {}
    for name, extra in ALSO_LEGS:
        cmd = [sys.executable, os.path.abspath(__file__), "--gpus", "1", "--steps", str(args.steps), "--warmup", str(args.warmup),
               "--precision", args.precision, "--brief"] + extra
        t0 = time.perf_counter()
        try:
            r = subprocess.run(cmd, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, timeout=600)
            line = [ln for ln in r.stdout.decode(errors="replace").splitlines() if ln.startswith("{")]
            j = json.loads(line[-1]) if (r.returncode == 0 and line) else None
        except (subprocess.TimeoutExpired, ValueError) as e:
            r, j = None, None
            out[name] = {"error": "%s: %s" % (type(e).__name__, e)}
            continue
        if j is None:
            out[name] = {"error": "rc %d: %s" % (r.returncode, r.stderr.decode(errors="replace")[-300:])}
            continue
        out[name] = {"workload": j["config"]["workload"], "ms_per_step": j["ms_per_step"], "frames_per_s": j["value"],
                     "launch": j["config"]["launch"], "launch_calibration_ms": j["config"].get("launch_calibration_ms"),
                     "steps": j["steps"], "warmup": j["warmup"], "losses_last_step": j.get("losses_last_step"),
                     "wall_s": round(time.perf_counter() - t0, 1)}
    return out


def build_model(asr_amd, dev, dropout, train):
    import torch
    torch.manual_seed(0)
    if CFG.get("cif"):
        a = argparse.Namespace(d_input=CFG["d_input"], LFR_m=1, d_model=CFG["d_model"], n_conv_layers=CFG["n_conv_layers"],
                               n_layers_enc=CFG["n_layers_enc"], n_head=CFG["n_head"], d_inner=CFG["d_inner"], dropout=dropout,
                               sos_id=CFG["sos_id"], eos_id=CFG["eos_id"], vocab_size=CFG["vocab_size"], n_layers_dec=CFG["n_layers_dec"],
                               spec_aug_cfg=None, d_assigner_hidden=CFG["d_model"], w_context=3, n_assigner_layers=3)
        model = asr_amd.CIF_Model.create_model(a).to(dev)
        return model.train() if train else model.eval()
    d_in = CFG["d_model"] if CFG["n_conv_layers"] else CFG["d_input"]
    enc = asr_amd.Encoder(d_in, CFG["n_layers_enc"], CFG["n_head"], CFG["d_model"], CFG["d_inner"], dropout=dropout)
    dec = asr_amd.Decoder(CFG["sos_id"], CFG["eos_id"], CFG["vocab_size"], CFG["n_layers_dec"], CFG["n_head"], CFG["d_model"],
                          CFG["d_inner"], dropout=dropout)
    if CFG["n_conv_layers"]:
        conv = asr_amd.Conv2dSubsample(CFG["d_input"], CFG["d_model"], n_layers=CFG["n_conv_layers"])
        model = asr_amd.Conv_CTC_Transformer(conv, enc, dec).to(dev)
    else:
        model = asr_amd.CTC_Transformer(enc, dec).to(dev)
    return model.train() if train else model.eval()


def make_batch(dev, seed, ragged=False):
    import torch
    g = torch.Generator().manual_seed(seed)
    B, T, U = CFG["B"], CFG["T"], CFG["U"]
    x = torch.randn(B, T, CFG["d_input"], generator=g)
    tg = torch.randint(4, CFG["vocab_size"] - 1, (B, U), generator=g)
    if ragged:
        lens = torch.randint(T // 2, T + 1, (B,), generator=g)
        lens[0] = T
        ul = torch.randint(U // 2, U + 1, (B,), generator=g)
        ul[0] = U
        tg = tg * (torch.arange(U)[None, :] < ul[:, None])
    else:
        lens = torch.full((B,), T, dtype=torch.int64)
    return x.to(dev), lens.to(dev), tg.to(dev)


def cpu_baseline(model, x, lens, tg, dropout, train):
    """Stock PyTorch CPU, the reference's op sequence (oracle/torch_cpu_ref.py, pinned on the reference's own outputs in
    tests/test_oracle_golden.py): forward + joint loss + backward on a bounded sample of this batch, all host cores; plus
    F.ctc_loss forward + backward alone at the batch's (B, T, U, V).  Bounded to ~10-30 s: the sample shrinks when the host is
    small (memory: the reference keeps [h*B, L, L] attention maps of every layer for autograd) or slow."""
    import torch
    from oracle import torch_cpu_ref as R
    avail_cores = os.cpu_count() or 1
    try:
        avail_cores = min(avail_cores, len(os.sched_getaffinity(0)))
    except AttributeError:
        pass
    try:      # cgroup v2 CPU quota of the container
        q, p = open("/sys/fs/cgroup/cpu.max").read().split()
        if q != "max":
            avail_cores = max(1, min(avail_cores, int(float(q) / float(p))))
    except (OSError, ValueError):
        pass
    # threads: every core the process may use, unless fewer run a GEMM faster (a pod can see 256 logical CPUs and still be given a
    # fraction of them: 256 threads then thrash - measured 223 s for what 8 cores do in 2 s)
    cands = sorted({c for c in (avail_cores, 128, 64, 32, 16, 8) if c <= avail_cores}, reverse=True)
    a = torch.randn(1536, 1536)
    best = None
    for c in cands:
        torch.set_num_threads(c)
        a @ a
        t0 = time.perf_counter()
        for _ in range(3):
            a @ a
        dt = time.perf_counter() - t0
        if best is None or dt < 0.8 * best[0]:      # fewer threads only when clearly faster
            best = (dt, c)
    cores = best[1]
    torch.set_num_threads(cores)
    try:
        import psutil
        avail = psutil.virtual_memory().available / 2 ** 30
    except Exception:
        avail = 32.0
    L = CFG["T"] // 4 if CFG["n_conv_layers"] else CFG["T"]
    per_utt = 1.0 + CFG["n_layers_enc"] * CFG["n_head"] * L * L * 4 * 3.2 / 2 ** 30     # ~GiB autograd keeps per utterance
    n_utt = CFG["B"]
    while n_utt > 2 and n_utt * per_utt > 0.45 * avail:
        n_utt //= 2
    model_name = ""
    try:
        for line in open("/proc/cpuinfo"):
            if line.startswith("model name"):
                model_name = line.split(":", 1)[1].strip()
                break
    except OSError:
        pass
    sd = R.leaves({k: v for k, v in model.state_dict().items()}, requires_grad=True)
    cfg = dict(n_layers_enc=CFG["n_layers_enc"], n_layers_dec=CFG["n_layers_dec"], n_head=CFG["n_head"], sos_id=CFG["sos_id"],
               eos_id=CFG["eos_id"])
    xs, ls, ts = x.cpu(), lens.cpu(), tg.cpu()

    def one(n):
        for t in sd.values():
            t.grad = None
        t0 = time.perf_counter()
        R.joint_step(sd, xs[:n], ls[:n], ts[:n], cfg, conv_layers=CFG["n_conv_layers"], p=dropout if train else 0.0, train=train,
                     backward=train)
        return time.perf_counter() - t0

    t_probe = one(2)                       # warm-up (thread pool, allocator) and a cost probe: 2 utterances
    # 1 warm-up + >= 3 timed iterations.  The sample is sized so that ONE iteration takes ~6 s (the whole leg ~25-30 s:
    # the contract bounds the CPU work of a default run); the full batch of 32 x 1000 frames takes ~25 s per iteration on these hosts.
    while n_utt > 2 and t_probe * n_utt / 2 > 6.5:
        n_utt //= 2
    if t_probe > 20.0:                     # a very slow host: the probe is the measurement
        iters, dt, dts = 1, t_probe, [t_probe]
    else:
        if n_utt > 2:
            one(n_utt)                     # warm-up at the timed size
        iters = 3
        dts = [one(n_utt) for _ in range(iters)]
        dt = sum(dts) / len(dts)
    ctc_iters = 20 if t_probe < 20.0 else 3
    ctc_ms = R.ctc_op(CFG["B"], L if CFG["n_conv_layers"] else CFG["T"], CFG["U"], CFG["vocab_size"], ctc_iters)
    what = "forward + loss + backward (train mode, dropout %g)" % dropout if train else "eval-mode forward + loss"
    return dict(value=round(n_utt * CFG["T"] / dt, 1), unit="frames/s", cores=cores, kind="port", impl="torch-cpu",
                port_of="stock PyTorch CPU ops composed in the reference's op order (oracle/torch_cpu_ref.py, pinned on the reference's "
                        "outputs and gradients) - the baseline SURVEY 8(d) prescribes; the reference's own files do not travel to this box",
                cpu_model=model_name, ms_per_step=round(dt * 1e3, 1), iterations_s=[round(v, 2) for v in dts], ctc_cpu_ms=round(ctc_ms, 2),
                threads={"used": cores, "os_cpu_count": os.cpu_count() or 1, "usable": avail_cores,
                         "why": "the fastest of {usable, 128, 64, 32, 16, 8} on a 1536^3 matmul probe: these pods show every host CPU but "
                                "schedule a fraction of them, and torch with os.cpu_count() threads then runs ~20x slower"},
                sample="%d of the batch's %d utterances (T=%d; one iteration of the full batch takes ~%.0f s here and the contract bounds the "
                       "CPU leg of a default run to ~30 s), %s, stock torch %s CPU ops in the reference's op order (oracle/torch_cpu_ref.py), "
                       "%d threads, 1 warm-up + %d timed iterations of %.2f s (mean); ctc_cpu_ms = F.log_softmax + F.ctc_loss forward + "
                       "backward at (B=%d, T=%d, U=%d, V=%d), mean of %d iterations" % (
                           n_utt, CFG["B"], CFG["T"], dt * CFG["B"] / n_utt, what, torch.__version__, cores, iters, dt, CFG["B"],
                           L if CFG["n_conv_layers"] else CFG["T"], CFG["U"], CFG["vocab_size"], ctc_iters))


def oracle_parity(asr_amd, model, x, lens, tg, n_utt=2):
    """max |GPU - numpy oracle| of both logit tensors on the first n_utt utterances (eval mode); asserted in tests/test_gpu_fullsize.py."""
    import numpy as np
    import torch
    from oracle import asr_oracle as O
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    cfg = dict(n_layers_enc=CFG["n_layers_enc"], n_layers_dec=CFG["n_layers_dec"], n_head=CFG["n_head"], sos_id=CFG["sos_id"],
               eos_id=CFG["eos_id"], n_conv_layers=CFG["n_conv_layers"])
    xs, ls, ts = x[:n_utt].cpu().numpy(), lens[:n_utt].cpu().numpy(), tg[:n_utt].cpu().numpy()
    was_training = model.training
    model.eval()
    with torch.no_grad():
        if CFG.get("cif"):
            # CIF_Model (cif_model.py:24-55) with a recorded noise vector; also the integrate-and-fire boundaries (north_star: "CIF firing
            # boundaries"): the frames at which the accumulated weight crosses the threshold must be the oracle's, exactly
            noise = torch.rand(n_utt, generator=torch.Generator().manual_seed(1))
            cfg.update(n_assigner_layers=3, w_context=3)
            ref_ctc, _, _, _, ref_logits, _, _, ref_fire = O.cif_model_forward(sd, xs, ls, ts, cfg, noise.numpy())
            cl, _, _, _, lg = model(x[:n_utt], lens[:n_utt], tg[:n_utt], noise=noise.to(x.device))
        elif CFG["n_conv_layers"]:
            ref_ctc, _, ref_logits = O.conv_ctc_transformer_forward(sd, xs, ls, ts, cfg)[:3]
            cl, _, lg, _ = model(x[:n_utt], lens[:n_utt], tg[:n_utt])
        else:
            _, ref_ctc, (ref_logits, _), _ = O.ctc_transformer_forward(sd, xs, ls, ts, cfg)
            _, cl, (lg, _) = model(x[:n_utt], lens[:n_utt], tg[:n_utt])
    model.train(was_training)
    return {"ctc_logits": float(np.abs(cl.float().cpu().numpy() - ref_ctc).max()),
            "logits": float(np.abs(lg.float().cpu().numpy() - ref_logits).max()), "utterances": n_utt}


# ---- kernel families: which device kernel an op name runs on (for the dominant-KERNEL pick and the PMC lookup) -------------------
def family(op_name):
    f = op_name.split("[", 1)[0]
    return "gemm_tn" if f == "gemm_tn_group" else f        # (a grouped launch = up to 8 weight gradients in one launch pair: same kernels)


def pmc_traffic(fam_kernel, prof_dir):
    """HBM bytes per launch of a device kernel from the committed PMC summary (separate rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE
    passes of this same bench; FETCH doubled per MI355X_MICROARCH.md §HBM), launch-weighted over its grid sizes.  None when unknown."""
    path = os.path.join(prof_dir, "pmc_traffic_train_s1.json")
    if not os.path.exists(path):
        return None
    pmc = json.load(open(path))
    tot, n = 0.0, 0
    for k, v in pmc.items():
        if k.split("|", 1)[0].split("<", 1)[0] == fam_kernel:
            tot += v["hbm_bytes"] * v["launches"]
            n += v["launches"]
    return int(tot / n) if n else None


FAMILY_KERNEL = {"vocab_proj_ctc": "vocab_proj_ctc_kernel", "ffn_fwd": "ffn_fwd2_kernel", "ffn_bwd": "ffn_bwd_kernel", "gemm_tn": "gemm_tn_v2_kernel", "gemm_nt": "gemm_nt_glds_kernel", "gemm_nn": "gemm_nn_tr_kernel",
                 "attention_fwd": "attn_fwd_bf16_v4a_kernel", "attention_bwd_dq": "attn_bwd_dq_v4_kernel", "attention_bwd_dkv": "attn_bwd_dkv_v4_kernel",
                 "add_layernorm": "add_layernorm_fwd_kernel", "add_layernorm_bwd": "add_layernorm_bwd_kernel",
                 "ctc_loss_fwd": "ctc_fused_fwd_kernel", "ctc_loss_fwd_table": "ctc_mitm_kernel", "ctc_loss_bwd": "ctc_grad_bf16_kernel", "proj_heads": "proj_heads_rows_kernel"}
# every device kernel an op family launches, as rocprofv3's kernel_stats.csv names them (substring match): what tools/roofline_from_csv.py
# sums to recompute a family's in-step rate from profiles/rN/bench_train_kernel_stats.csv
FAMILY_CSV_KERNELS = {"gemm_tn": ["gemm_tn_v2_kernel", "gemm_tn_v2_group_kernel", "tn_reduce_kernel", "tn_reduce_group_kernel", "gemm_tn_kernel"],
                      "ffn_fwd": ["ffn_fwd2_kernel"], "ffn_bwd": ["ffn_bwd_kernel"], "attention_fwd": ["attn_fwd_bf16_v4a_kernel", "attn_fwd_bf16_v2_kernel"],
                      "attention_bwd_dq": ["attn_bwd_dq_v4_kernel", "attn_bwd_dq_kernel"], "attention_bwd_dkv": ["attn_bwd_dkv_v4_kernel", "attn_bwd_dkv_kernel"],
                      "vocab_proj_ctc": ["vocab_proj_ctc_kernel"], "ctc_loss_fwd": ["ctc_fused_fwd_kernel", "ctc_mitm_kernel"], "ctc_loss_fwd_table": ["ctc_mitm_kernel"],
                      "ctc_loss_bwd": ["ctc_grad_bf16_kernel", "ctc_grad_kernel", "ctc_mitm_kernel"]}


def main():
    args = parse_args()
    world_env = os.environ.get("WORLD_SIZE")
    if args.gpus > 1 and world_env is None:
        sys.exit(spawn_ranks(args))
    also = None
    if (world_env is None and args.gpus == 1 and not args.brief and not args.no_also and args.model == "s1" and args.mode == "train"
            and not args.ragged):
        also = run_also_legs(args)        # children first: nothing in this process has touched the GPU yet
    import torch

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(world_env or "1")
    if world != args.gpus:
        raise SystemExit("bench.py: --gpus %d but WORLD_SIZE=%d (launch with --nproc-per-node %d)" % (args.gpus, world, args.gpus))
    if args.model in ("s2", "cif"):
        CFG["n_conv_layers"] = 2
    if args.model == "aishell":
        # the width the reference ships (egs/aishell/recipes/transformer.sh:18-53: d_model 512, 8 heads, 6 + 6 layers, d_inner 2048, LFR 4 / 3
        # stacking of 80-d fbank: 320-d frames at a third of the rate), same 32 x 1000 raw frames per GPU
        CFG.update(aishell=True, d_model=512, n_head=8, n_layers_enc=6, n_layers_dec=6, d_input=320, T=334, raw_T=1000)    # (frames/s counts the 80-d frames)
    if args.model == "cif":
        CFG["cif"] = True
        args.beam = args.beam or 5
        if args.mode == "fwd":
            raise SystemExit("bench.py: --model cif runs --mode train or --mode decode")
    if world > 1:
        import torch.distributed as dist
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        # nccl (= RCCL over xGMI) is the product backend; ASR_AMD_DIST_BACKEND=gloo + ASR_AMD_DEVICE=0 lets several ranks share one
        # GPU so the data-parallel step can be exercised on a single-GPU box (test rig only)
        dist.init_process_group(os.environ.get("ASR_AMD_DIST_BACKEND", "nccl"))
    dev_index = int(os.environ.get("ASR_AMD_DEVICE", local_rank))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)

    import asr_amd
    from asr_amd import ops
    asr_amd.set_precision(args.precision)
    train = args.mode == "train"
    model = build_model(asr_amd, dev, args.dropout, train=train)
    asr_amd.manual_seed(1234 + rank)       # dropout masks: reproducible, different on every rank
    x, lens, tg = make_batch(dev, seed=rank, ragged=args.ragged)

    def ctc_standalone_ms():
        """the CTC forward as SURVEY 8(d) row 2 prices it: the stand-alone op on resident fp32 logits of the workload's shape, 30 launches
        back-to-back (it reads the logits in full every time)"""
        Lc_ = CFG["T"] // 4 if CFG["n_conv_layers"] else CFG["T"]
        g = torch.Generator().manual_seed(0)
        lg_ = torch.randn(CFG["B"], Lc_, CFG["vocab_size"], generator=g).to(dev)
        tg_ = torch.randint(1, CFG["vocab_size"] - 1, (CFG["B"], CFG["U"]), generator=g).to(dev)
        il_ = torch.full((CFG["B"],), Lc_, dtype=torch.int32, device=dev)
        reps = []
        for _ in range(3):
            for _ in range(5):
                ops.ctc_loss_fwd(lg_, il_, tg_)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                ops.ctc_loss_fwd(lg_, il_, tg_)
            e1.record()
            torch.cuda.synchronize()
            reps.append(e0.elapsed_time(e1) / 30)
        del lg_
        torch.cuda.empty_cache()
        return sorted(reps)[1]      # the median of the three groups

    # before anything else has run on the chip (the same measurement is repeated after the run: the alpha / beta chains are dependent
    # VALU chains, i.e. clock-bound, and the chip's clock under ~a minute of sustained MFMA load is lower than that of an idle one)
    ctc_iso_first = ctc_standalone_ms() if (rank == 0 and not args.brief and args.mode != "decode" and not CFG.get("cif")) else None

    trainer = (asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1, **({"lambda_qua": 0.001} if CFG.get("cif") else {}))
               if train else None)
    if trainer is not None and args.side_budget >= 0:
        trainer.side_budget = args.side_budget
    for kv in (args.trainer_set if trainer is not None else []):
        name, val = kv.split("=")
        assert hasattr(trainer, name), "Trainer has no attribute %r" % name
        setattr(trainer, name, type(getattr(trainer, name))(int(val)))
    use_graph = args.graph == 1 and trainer is not None
    auto_graph = args.graph < 0 and trainer is not None


    n_steps = [0]

    def step():
        n_steps[0] += 1
        if trainer is not None:
            if use_graph:
                return trainer.step_graphed(x, lens, tg, max_target_len=CFG["U"])
            if auto_graph:
                return trainer.step_auto(x, lens, tg, max_target_len=CFG["U"])
            return trainer.step(x, lens, tg, max_target_len=CFG["U"])   # the loader knows its target lengths: no host sync in the step
        if args.mode == "decode" and CFG.get("cif"):
            # the reference's CIF inference (cif_model.py:108-131) for the whole padded batch: one batched beam search; `target_num` = U
            # tokens per utterance (the random-init assigner would otherwise fire on about every second frame)
            res = model.batch_recognize(x, lens, args.beam, 1, target_num=CFG["U"])
            t_ = torch.tensor(sum(ls[0] - 1 for ys, ls in res) / float(x.shape[0]))
            return t_, t_
        if args.mode == "decode":
            with torch.no_grad():
                from asr_amd.modules import _act
                if CFG["n_conv_layers"]:
                    conv, l = model.conv_encoder._impl(x, lens)
                    enc = model.encoder._impl(conv, l)
                else:
                    l = ops.as_i32(lens, dev)
                    enc = model.encoder._impl(_act(x), l)
                if args.beam > 0:
                    preds, n_dec, _ = model.decoder.batch_beam_decode(enc.view3(), l, args.beam, args.decode_len)
                else:
                    preds, n_dec, _ = model.decoder.batch_decode(enc.view3(), l, args.decode_len)
                toks, n_ctc = asr_amd.ctc_greedy_decode(model._ctc_logits(enc).view(enc.B, enc.L, -1), l)
            return n_dec.float().mean(), n_ctc.float().mean()
        with torch.no_grad():
            out = model(x, lens, tg)
            if CFG["n_conv_layers"]:
                ctc_logits, l, logits, teos = out
            else:
                l, ctc_logits, (logits, teos) = out
            ctc, ce = asr_amd.cal_ctc_ce_loss(ctc_logits, l, logits, teos, 0.1)
        return ctc, ce

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # initialisation, before the W warmup steps of the contract: the first steps of a process grow the caching allocator's pools,
    # load code objects, create the side streams / event pool, capture the step's HIP graph and (N > 1) set up the RCCL communicators
    for _ in range(3):       # (with --graph -1 the first of these also runs step_auto's calibration: 2 + 4 eager steps, capture, 4 replays)
        step()
    torch.cuda.synchronize()
    # a fresh box keeps speeding up for a while (clocks, page tables, the allocator's pools): groups of 5 steps until two consecutive
    # groups agree to 2 % (at most 8 groups), THEN the contract's W warm-up steps and the K timed ones
    def timed_group():
        t0 = time.perf_counter()
        for _ in range(5):
            step()
        torch.cuda.synchronize()
        return (time.perf_counter() - t0) / 5 * 1e3
    settle = settle_groups(timed_group, world, dev)
    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        out = step()
    torch.cuda.synchronize()
    barrier()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    rccl_ranks = None
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())
        ones = torch.ones(1, device=dev)
        torch.distributed.all_reduce(ones)          # every rank counted by the collective itself, not by the environment
        rccl_ranks = {"ranks": int(ones.item()), "world_size": torch.distributed.get_world_size(),
                      "backend": torch.distributed.get_backend()}
    graphed = bool(trainer is not None and trainer.graph_active() and (use_graph or (auto_graph and trainer.launch_mode == "graph")))
    launch_timing = trainer.launch_timing if trainer is not None else None
    losses = [float(v) for v in out]

    # ---- live per-kernel timing over a second, identical run of the timed region (events add launch overhead, so the
    # headline value above is measured without them; eager launches, one stream: per-op durations are then uncontended) ----
    if args.brief:
        if rank == 0:
            frames = world * CFG["B"] * CFG.get("raw_T", CFG["T"]) * args.steps
            print(json.dumps({"metric": "fbank frames/sec", "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world,
                              "steps": args.steps, "warmup": args.warmup, "ms_per_step": round(dt / args.steps * 1e3, 3),
                              "losses_last_step": losses, "settle_ms": [round(v, 3) for v in settle],
                              "config": {"workload": workload_name(args, train), "launch": launch_name(args, graphed, trainer),
                                         "launch_calibration_ms": launch_timing}}))
        if world > 1:
            torch.distributed.destroy_process_group()
        return
    # Two more passes over the same steps, eager, every C-ABI call bracketed by HIP timing events on the stream it is launched on:
    #  (1) IN STEP: the step exactly as it was timed above - CTC branch, weight gradients and dropout-mask hashing on their side
    #      streams - so every bracket holds the kernel's duration WITH its neighbours on the chip (what rocprofv3's kernel_stats.csv of
    #      this command averages; tools/roofline_from_csv.py recomputes roofline.frac_in_step from that file);
    #  (2) ALONE: the same launches, same kernels, all on one stream - the side branches inlined, the masks hashed in line - so
    #      every bracket holds the kernel by itself.  (Round 4 ran this pass with the mask stream still live - twelve 60 us hashing
    #      launches landing inside ffn_fwd's brackets: 222 us for a 97-115 us kernel - and with the streaming CTC forward instead of
    #      the projection + lse form the step runs.)
    prof_in_step = None
    if trainer is not None:
        use_graph = auto_graph = False
        ops.profile_start()
        for _ in range(args.steps):
            step()
        prof_in_step = ops.profile_stop()
        from asr_amd import modules as _modules
        if args.per_op == "both":
            trainer.wgrad_stream = False
            trainer.side_inline = True              # the CTC branch's own kernels (vocab_proj_ctc, recursion, gradient) on the launch stream
            _modules._MASK_PREFETCH = False
    if prof_in_step is not None and args.per_op == "in_step":
        prof = prof_in_step
    else:
        ops.profile_start()
        for _ in range(args.steps):
            step()
        prof = ops.profile_stop()

    if rank == 0:
        frames = world * CFG["B"] * CFG.get("raw_T", CFG["T"]) * args.steps
        kernels, fams = [], {}
        hbm_ops = ("add_layernorm", "ctc_loss", "ce_loss", "cif_", "adam")
        for name, r in prof.items():
            per_ms = r["ms"] / r["calls"]
            hbm = name.startswith(hbm_ops)
            ach = (r["work"] / r["calls"]) / (per_ms * 1e-3) / (1e9 if hbm else 1e12)
            kernels.append(dict(name=name, calls_per_step=r["calls"] / args.steps, ms_per_call=round(per_ms, 4),
                                ms_per_step=round(r["ms"] / args.steps, 3), bound="hbm" if hbm else "mfma",
                                achieved=round(ach, 2), unit="GB/s" if hbm else "TFLOP/s"))
            f = fams.setdefault(family(name), dict(ms=0.0, work=0.0, calls=0, hbm=hbm))
            f["ms"] += r["ms"]
            f["work"] += r["work"]
            f["calls"] += r["calls"]
        kernels.sort(key=lambda k: -k["ms_per_step"])
        fams_in = {}
        for name, r in (prof_in_step or {}).items():
            f = fams_in.setdefault(family(name), dict(ms=0.0, work=0.0, calls=0, hbm=name.startswith(hbm_ops)))
            f["ms"] += r["ms"]
            f["work"] += r["work"]
            f["calls"] += r["calls"]
        # dominant KERNEL = the family (all shapes of one device kernel) with the most time in the step
        dom_name, dom = max(((n, f) for n, f in fams.items() if f["work"] > 0), key=lambda nf: nf[1]["ms"])
        ach = dom["work"] / (dom["ms"] * 1e-3) / (1e9 if dom["hbm"] else 1e12)
        peak = PEAK_HBM_GBS if dom["hbm"] else PEAK_MFMA_BF16_TFLOPS
        dom_in = fams_in.get(dom_name)
        ach_in = (dom_in["work"] / (dom_in["ms"] * 1e-3) / (1e9 if dom["hbm"] else 1e12)) if dom_in and dom_in["ms"] > 0 else None
        prof_dir = next((d for d in (os.path.join(ROOT, "profiles", r) for r in ("r6", "r5", "r4", "r3", "r2", "r1"))
                         if os.path.exists(os.path.join(d, "pmc_traffic_train_s1.json"))), os.path.join(ROOT, "profiles", "r6"))
        traffic = pmc_traffic(FAMILY_KERNEL.get(dom_name, dom_name), prof_dir)
        roofline = dict(kernel="%s (%s, all shapes)" % (dom_name, FAMILY_KERNEL.get(dom_name, dom_name)),
                        bound="hbm" if dom["hbm"] else "mfma", achieved=round(ach, 2), peak=peak,
                        unit="GB/s" if dom["hbm"] else "TFLOP/s", frac=round(ach / peak, 4),
                        frac_in_step=(round(ach_in / peak, 4) if ach_in else None),
                        achieved_in_step=(round(ach_in, 2) if ach_in else None),
                        ms_per_step_in_step=(round(dom_in["ms"] / args.steps, 3) if dom_in else None),
                        algorithmic_work_per_step=dom["work"] / args.steps,
                        csv_kernels=FAMILY_CSV_KERNELS.get(dom_name, [FAMILY_KERNEL.get(dom_name, dom_name)]),
                        launches_per_step=dom["calls"] / args.steps, ms_per_step=round(dom["ms"] / args.steps, 3),
                        avg_launch_us=round(dom["ms"] / dom["calls"] * 1e3, 2),
                        traffic=traffic,
                        traffic_source=(None if traffic is None else
                                        "%s/pmc_traffic_train_s1.json: a committed profile of this same command (rocprofv3 --pmc FETCH_SIZE and "
                                        "--pmc WRITE_SIZE in separate passes), NOT measured in this run" % os.path.relpath(prof_dir, ROOT)),
                        note="dominant kernel = the op family (one device kernel, all shapes summed) with the most time per step; "
                             "achieved / frac = algorithmic FLOPs (or bytes) of all its launches / their summed duration from HIP events on the "
                             "launch stream, in an eager pass of the same steps with every side branch inlined on one stream (each kernel "
                             "alone on the chip); achieved_in_step / frac_in_step = the same from an eager pass WITH the step's side streams "
                             "(each kernel beside its neighbours, as `value` is measured) - recompute it from rocprofv3's kernel_stats.csv of "
                             "this command: algorithmic_work_per_step x steps / total duration of csv_kernels (tools/roofline_from_csv.py); "
                             "traffic = mean HBM bytes per launch from separate rocprofv3 "
                             "--pmc FETCH_SIZE / WRITE_SIZE passes (%s/pmc_traffic_train_s1.json; FETCH doubled per the gfx950 "
                             "correction)" % os.path.relpath(prof_dir, ROOT))
        if dom_name == "gemm_tn" and args.mode == "train":
            # the dominant family's heaviest shape (the FFN weight gradients: 2 x 12 of the 88 calls, 60 % of the family's flops) back to
            # back, 30 launches inside ONE event bracket: a per-op bracket costs 15-30 us of event handling on this stack, which the
            # family number above carries 88 times per step
            Mr, dff = CFG["B"] * (CFG["T"] // 4 if CFG["n_conv_layers"] else CFG["T"]), CFG["d_inner"]
            g_ = torch.Generator(device="cpu").manual_seed(1)
            da = torch.randn(Mr, 256, generator=g_).to(dev).bfloat16()
            xb_ = torch.randn(Mr, dff, generator=g_).to(dev).bfloat16()
            out_ = torch.empty(256, dff, device=dev)
            for _ in range(3):
                ops.gemm_tn(da, xb_, out=out_, max_wgs=256)
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(30):
                ops.gemm_tn(da, xb_, out=out_, max_wgs=256)
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            tf = 2.0 * Mr * 256 * dff / (us * 1e-6) / 1e12
            roofline["back_to_back"] = {"shape": [Mr, 256, dff], "us": round(us, 2), "achieved": round(tf, 1), "unit": "TFLOP/s",
                                        "frac": round(tf / PEAK_MFMA_BF16_TFLOPS, 4),
                                        "what": "the family's heaviest shape (FFN weight gradient, both launches of the op), 30 calls inside one event bracket"}
            del da, xb_, out_
        ctc_k = [k for k in kernels if k["name"].startswith("ctc_loss_fwd")]
        ctc_b = [k for k in kernels if k["name"].startswith("ctc_loss_bwd")]
        vp_k = [k for k in kernels if k["name"].startswith("vocab_proj_ctc")]
        what, mname = what_name(args, train), model_name()
        Lc = CFG["T"] // 4 if CFG["n_conv_layers"] else CFG["T"]
        ctc_iso = ctc_standalone_ms() if (ctc_k and args.mode != "decode") else None
        ctc_block = None
        if ctc_k:
            logit_bytes = 4.0 * CFG["B"] * Lc * CFG["vocab_size"]

            def pick(profd, prefix):
                r = [(n, v) for n, v in (profd or {}).items() if n.startswith(prefix)]
                return (sum(v["ms"] for _, v in r) / max(1, sum(v["calls"] for _, v in r))) if r else None
            branch = {}
            for tag, profd in (("alone", prof), ("in_step", prof_in_step)):
                vp, fw, bw = pick(profd, "vocab_proj_ctc"), pick(profd, "ctc_loss_fwd"), pick(profd, "ctc_loss_bwd")
                if fw is None:
                    continue
                branch[tag] = {"vocab_proj_ctc_ms": None if vp is None else round(vp, 4), "ctc_recursion_ms": round(fw, 4),
                               "ctc_bwd_ms": None if bw is None else round(bw, 4),
                               "branch_fwd_bwd_ms": (round((vp or 0.0) + fw + (bw or 0.0), 4))}
            ctc_block = {
                "form_in_step": ("ctc_fc's projection writes fp16 logits, their row log-sum-exp AND the CTC table rows (the ~52 logits per frame the "
                                 "utterance's extended label sequence needs, picked out of the fp32 accumulators as they pass through LDS: "
                                 "vocab_proj_ctc_kernel); the CTC forward is then the alpha / beta recursion on that table and never touches the "
                                 "logits (ctc_mitm_kernel); the backward streams the fp16 logits once and writes the bf16 gradient image "
                                 "(ctc_mitm_kernel + ctc_grad_bf16_kernel) - all on the trainer's side stream beside the decoder" if vp_k else
                                 "one launch: persistent pass workgroups stream the logits, the recursion waves consume the table rows as they arrive"),
                "branch_ms_per_call": branch,
                "ms_per_step_fwd": ctc_k[0]["ms_per_step"], "ms_per_step_bwd": (ctc_b[0]["ms_per_step"] if ctc_b else None),
                "fwd_ms_standalone": (round(ctc_iso_first, 4) if ctc_iso_first else None),
                "fwd_GBps_standalone": (round(logit_bytes / (ctc_iso_first * 1e-3) / 1e9, 1) if ctc_iso_first else None),
                "fwd_frac_of_hbm_peak_standalone": (round(logit_bytes / (ctc_iso_first * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if ctc_iso_first else None),
                "fwd_ms_standalone_after_run": (round(ctc_iso, 4) if ctc_iso else None),
                "fwd_frac_of_hbm_peak_standalone_after_run": (round(logit_bytes / (ctc_iso * 1e-3) / 1e9 / PEAK_HBM_GBS, 4) if ctc_iso else None),
                "note": "branch_ms_per_call: the CTC branch's ops as the timed step runs them (HIP events on their launch stream), `in_step` = on "
                        "the side stream beside the decoder, `alone` = the same launches inlined on one stream; *_standalone: the STREAMING "
                        "form of the op by itself (asr_ctc_loss_mean_fwd on resident fp32 logits of the same shape, which it reads in full: one "
                        "launch, median of 3 x 30 back-to-back) - the form every caller without a precomputed row lse gets, and the one the north-star's "
                        "HBM-roofline fraction is quoted on (algorithmic bytes = B x L x V x 4, SURVEY 8(d)); measured twice: as the first thing "
                        "this process runs on the chip, and again (`_after_run`) behind the training steps and per-op passes"}
        # ---- the north-star's own numbers as FLAT scalars inside `roofline` (the driver's record keeps scalars of the objects it knows and
        # drops nested values): encoder self-attention forward against the bf16 MFMA peak, the CTC alpha/beta kernel against the HBM peak
        def _op(profd, prefix):
            r = [v for n, v in (profd or {}).items() if n.startswith(prefix)]
            calls = sum(v["calls"] for v in r)
            return (sum(v["ms"] for v in r) / calls, sum(v["work"] for v in r) / calls) if calls else (None, None)
        for tag, profd in (("", prof), ("_in_step", prof_in_step)):
            ms_, work_ = _op(profd, "attention_fwd[B%d h%d %dx%d]" % (CFG["B"], CFG["n_head"], Lc, Lc))
            if ms_:
                roofline["attn_fwd_us" + tag] = round(ms_ * 1e3, 2)
                roofline["attn_fwd_frac" + tag] = round(work_ / (ms_ * 1e-3) / 1e12 / PEAK_MFMA_BF16_TFLOPS, 4)
        if "back_to_back" in roofline:
            roofline["back_to_back_frac"] = roofline["back_to_back"]["frac"]
            roofline["back_to_back_us"] = roofline["back_to_back"]["us"]
        if ctc_block is not None:
            lb = 4.0 * CFG["B"] * Lc * CFG["vocab_size"]
            # the streaming forward (one launch reads the fp32 logits once): MEDIAN of 3 x 30 back-to-back launches, before and after the run
            for tag, ms_ in (("", ctc_iso_first), ("_after_run", ctc_iso)):
                if ms_:
                    roofline["ctc_fwd_ms" + tag] = round(ms_, 4)
                    roofline["ctc_fwd_frac_hbm" + tag] = round(lb / (ms_ * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
            al = ctc_block["branch_ms_per_call"].get("alone")
            if al and al.get("ctc_bwd_ms") is not None:
                # the trainer's form, forward + backward, alone on the chip: projection epilogue aside, the recursion reads / writes the
                # tables (lp_ext, alpha: 2 x B x L x 128 x 4 per direction) and the gradient pass reads the fp16 logits + writes the bf16 image
                fb_ms = al["ctc_recursion_ms"] + al["ctc_bwd_ms"]
                fb_bytes = CFG["B"] * Lc * (CFG["vocab_size"] * (2.0 + 2.0) + 4 * 128 * 4.0)
                roofline["ctc_fwdbwd_ms"] = round(fb_ms, 4)
                roofline["ctc_fwdbwd_frac_hbm"] = round(fb_bytes / (fb_ms * 1e-3) / 1e9 / PEAK_HBM_GBS, 4)
        result = {
            "metric": "fbank frames/sec (%s d%d h%d enc%d/dec%d, %s)" % (mname.split(":")[1].strip().split(" ")[0], CFG["d_model"], CFG["n_head"], CFG["n_layers_enc"], CFG["n_layers_dec"], what),
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic", "rccl_ranks": rccl_ranks,
            "config": {"workload": workload_name(args, train),
                       "global_batch": world * CFG["B"], "seq_len": CFG["T"], "parallelism": "dp%d" % world,
                       # flat scalars the driver's record keeps: ranks counted by an all-reduce of ones (not by the environment), and its backend
                       "rccl_ranks": (rccl_ranks["ranks"] if rccl_ranks else 1), "collective_backend": (rccl_ranks["backend"] if rccl_ranks else "none"),
                       "launch": launch_name(args, graphed, trainer),
                       "launch_calibration_ms": launch_timing,       # step_auto's own eager vs replay timing (4 steps each, during initialisation)
                       "settle_ms": [round(v, 3) for v in settle]},   # ms/step of the 5-step groups run until steady, before the W warm-up steps
            "losses_last_step": losses,
            # training steps this process executed in all (initialisation + launch calibration + settle groups + warm-up + timed + per-op
            # passes): the divisor for per-step figures taken from a rocprofv3 kernel_stats.csv of this command
            "steps_executed": (int(trainer.step_num) if trainer is not None else n_steps[0]),
            "roofline": roofline,
            "ctc": ctc_block,
            "kernels": kernels[:12], "op_ms_total": round(sum(k["ms_per_step"] for k in kernels), 3),
            "families": sorted(({"family": n, "ms_per_step": round(f["ms"] / args.steps, 3),
                                 "achieved": round(f["work"] / (f["ms"] * 1e-3) / (1e9 if f["hbm"] else 1e12), 1) if f["work"] else None,
                                 "unit": "GB/s" if f["hbm"] else "TFLOP/s"} for n, f in fams.items()), key=lambda d: -d["ms_per_step"])[:8],
            # the same table from the in-step pass (kernels beside their neighbours on the side streams): what kernel_stats.csv shows
            "families_in_step": sorted(({"family": n, "ms_per_step": round(f["ms"] / args.steps, 3),
                                         "achieved": round(f["work"] / (f["ms"] * 1e-3) / (1e9 if f["hbm"] else 1e12), 1) if f["work"] else None,
                                         "unit": "GB/s" if f["hbm"] else "TFLOP/s"} for n, f in fams_in.items()), key=lambda d: -d["ms_per_step"])[:8],
        }
        if also is not None:
            # the other workloads BASELINE.json names, measured by this same command (child processes that ran before this one touched the GPU)
            result["also"] = also
        if world == 1 and not args.no_cpu_baseline and args.mode != "decode" and not CFG.get("cif"):
            result["cpu_baseline"] = cpu_baseline(model, x, lens, tg, args.dropout, train)
            # sanity: the GPU result on the same utterances agrees with the numpy oracle (bf16 tolerance); not timed
            result["parity_vs_oracle_max_abs"] = oracle_parity(asr_amd, model, x, lens, tg, n_utt=2)
        # key order: the driver keeps the LAST 2 000 characters of stdout verbatim - prose (notes, sample descriptions) first, the
        # numbers the north-star is written around (roofline's scalars, the ctc block's) last
        def _prose_first(d):
            if not isinstance(d, dict):
                return d
            long_ = {k: v for k, v in d.items() if isinstance(v, str) and len(v) > 60}
            nested = {k: v for k, v in d.items() if isinstance(v, (dict, list)) and k not in long_}
            rest = {k: v for k, v in d.items() if k not in long_ and k not in nested}
            return {**long_, **nested, **rest}
        tail_keys = ("cpu_baseline", "parity_vs_oracle_max_abs", "ctc", "roofline")
        ordered = {k: v for k, v in result.items() if k not in tail_keys}
        for k in tail_keys:
            if k in result:
                ordered[k] = _prose_first(result[k])
        print(json.dumps(ordered))
    if world > 1:
        if trainer is not None:   # data-parallel invariant: every rank holds bit-identical parameters after the same steps
            chk = trainer.fp.flat.double().sum().reshape(1)
            lo, hi = chk.clone(), chk.clone()
            torch.distributed.all_reduce(lo, op=torch.distributed.ReduceOp.MIN)
            torch.distributed.all_reduce(hi, op=torch.distributed.ReduceOp.MAX)
            assert float(lo) == float(hi), "ranks diverged: parameter checksum %r vs %r" % (float(lo), float(hi))
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
