#!/usr/bin/env python3
"""bench.py — the hot path's headline metric on MI355X (contract: see the task statement / DESIGN.md §Measurement).

Workload (BASELINE.json configs[1], SURVEY.md §8d "S1"): CTC_Transformer (no conv front-end), d_model=256 h=4
(d_k=d_v=64) d_inner=2048 enc12/dec6, V=4234, per-GPU batch B=32 x T=1000 x 80-dim fbank (full-length
utterances), U=50, bf16 MFMA operands / fp32 accumulate.  One "step" = one pass of the hot path over one
synthetic batch: encoder + ctc_fc + decoder forward, joint CTC + label-smoothed-CE loss
[+ backward + RCCL gradient all-reduce + Adam when --train is given].
value = input fbank frames per second summed over all ranks (weak scaling: per-GPU batch fixed).

rank 0 prints ONE JSON line.  Extra objects: "roofline" (dominant kernel, timed live with HIP events on the launch
stream inside the timed region), "cpu_baseline" (the numpy oracle of the same forward+loss on a bounded sample of
the same batch, timed on this node's host cores; N=1 only), "kernels" (per-kernel live timings), "ctc" (the fused
CTC loss op alone).
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

import torch  # noqa: E402

PMC_FILE = os.path.join(ROOT, "profiles", "r1", "pmc_traffic_train_s1.json")   # rocprofv3 --pmc passes, tools/pmc_summary.py


def pmc_traffic(op_name):
    """HBM bytes per launch of the kernel(s) behind a bench op, from the committed PMC summary (separate rocprofv3 --pmc
    FETCH_SIZE / WRITE_SIZE passes of this same bench; FETCH doubled per MI355X_MICROARCH.md §HBM).  None when unknown."""
    if not os.path.exists(PMC_FILE):
        return None
    pmc = json.load(open(PMC_FILE))
    B, T, h = CFG["B"], CFG["T"], CFG["n_head"]
    tiles = lambda m, n: ((m + 127) // 128) * ((n + 127) // 128) * 256
    M = B * T
    table = {
        "attention_bwd_dq[B%d h%d %dx%d]" % (B, h, T, T): [["attn_bwd_dq_kernel|grid=%d" % (B * h * ((T + 127) // 128) * 256)]],
        "attention_bwd_dkv[B%d h%d %dx%d]" % (B, h, T, T): [["attn_bwd_dkv_kernel|grid=%d" % (B * h * ((T + 127) // 128) * 256)]],
        "attention_fwd[B%d h%d %dx%d]" % (B, h, T, T): [["attn_fwd_bf16_v2_kernel|grid=%d" % (B * h * ((T + 127) // 128) * 256)],
                                                       ["attn_fwd_bf16_kernel|grid=%d" % (B * h * ((T + 127) // 128) * 256)]],
        "ctc_loss_bwd[B%d L%d V%d U%d]" % (B, T, CFG["vocab_size"], CFG["U"] + 1): [["ctc_grad_kernel|grid=%d" % (64 * B * 256)]],
        "gemm_nn[%dx%dx%d]" % (M, CFG["d_inner"], CFG["d_model"]): [["gemm_nn_tr_kernel<mode80>|grid=%d" % tiles(M, CFG["d_inner"])],
                                                                    ["gemm_nn_tr_kernel<mode24>|grid=%d" % tiles(M, CFG["d_inner"])]],
        # (the LDS-DMA NT kernel is persistent: 512 workgroups; the compile-time epilogue mode tells the FFN's first GEMM apart)
        "gemm_nt[%dx%dx%d]" % (M, CFG["d_inner"], CFG["d_model"]): [["gemm_nt_glds_kernel<mode51>|grid=%d" % (512 * 256)],
                                                                    ["gemm_nt_glds_kernel<mode19>|grid=%d" % (512 * 256)]],
    }
    for keys in table.get(op_name, []):
        if all(k in pmc for k in keys):
            return int(sum(pmc[k]["hbm_bytes"] for k in keys))
    return None




PEAK_MFMA_BF16_TFLOPS = 2500.0   # dense bf16 MFMA peak, MI355X_MICROARCH.md "Chip-level parameters"
PEAK_HBM_GBS = 8000.0            # HBM3E spec peak, same table

CFG = dict(d_input=80, d_model=256, n_head=4, d_inner=2048, n_layers_enc=12, n_layers_dec=6, vocab_size=4234,
           sos_id=2, eos_id=3, B=32, T=1000, U=50)


def build_model(asr_amd, dev, dropout, train):
    torch.manual_seed(0)
    enc = asr_amd.Encoder(CFG["d_input"], CFG["n_layers_enc"], CFG["n_head"], CFG["d_model"], CFG["d_inner"], dropout=dropout)
    dec = asr_amd.Decoder(CFG["sos_id"], CFG["eos_id"], CFG["vocab_size"], CFG["n_layers_dec"], CFG["n_head"], CFG["d_model"],
                          CFG["d_inner"], dropout=dropout)
    model = asr_amd.CTC_Transformer(enc, dec).to(dev)
    return model.train() if train else model.eval()


def make_batch(dev, seed):
    g = torch.Generator().manual_seed(seed)
    x = torch.randn(CFG["B"], CFG["T"], CFG["d_input"], generator=g)
    lens = torch.full((CFG["B"],), CFG["T"], dtype=torch.int64)
    tg = torch.randint(4, CFG["vocab_size"] - 1, (CFG["B"], CFG["U"]), generator=g)
    return x.to(dev), lens.to(dev), tg.to(dev)


def cpu_baseline(model, x, lens, tg, n_utt=1):
    """numpy oracle (oracle/asr_oracle.py = CPU port of the reference's arithmetic) on the first n_utt utterances."""
    import numpy as np
    from oracle import asr_oracle as O
    sd = {k: v.detach().float().cpu().numpy() for k, v in model.state_dict().items()}
    cfg = dict(n_layers_enc=CFG["n_layers_enc"], n_layers_dec=CFG["n_layers_dec"], n_head=CFG["n_head"], sos_id=CFG["sos_id"],
               eos_id=CFG["eos_id"])
    xs, ls, ts = x[:n_utt].cpu().numpy(), lens[:n_utt].cpu().numpy(), tg[:n_utt].cpu().numpy()
    t0 = time.time()
    l, ctc_logits, (logits, teos), _ = O.ctc_transformer_forward(sd, xs, ls, ts, cfg)
    ctc, ce = O.cal_ctc_ce_loss(ctc_logits, l, logits, teos, 0.1)
    dt = time.time() - t0
    return dict(value=float(n_utt * CFG["T"] / dt), unit="frames/s", cores=os.cpu_count(), kind="port",
                sample="%d of the batch's %d utterances (T=%d), forward+loss, numpy fp32 oracle, %.1f s" % (
                    n_utt, CFG["B"], CFG["T"], dt)), float(ctc), float(ce), ctc_logits, logits


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=10)
    ap.add_argument("--warmup", type=int, default=3)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--precision", default="bf16")
    ap.add_argument("--mode", default="train", choices=["train", "fwd"],
                    help="train: forward+loss+backward+grad all-reduce+Adam in train mode (default); fwd: eval-mode forward+loss only")
    ap.add_argument("--dropout", type=float, default=0.1,
                    help="dropout rate of the training step (0.1 = every shipped config of the reference, egs/*/conf); ignored by --mode fwd")
    args = ap.parse_args()

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
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
    model = build_model(asr_amd, dev, args.dropout, train=(args.mode == "train"))
    asr_amd.manual_seed(1234 + rank)       # dropout masks: reproducible, different on every rank
    x, lens, tg = make_batch(dev, seed=rank)

    trainer = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1) if args.mode == "train" else None

    def step():
        if trainer is not None:
            return trainer.step(x, lens, tg, max_target_len=CFG["U"])   # the loader knows its target lengths: no host sync in the step
        with torch.no_grad():
            l, ctc_logits, (logits, teos) = model(x, lens, tg)
            ctc, ce = asr_amd.cal_ctc_ce_loss(ctc_logits, l, logits, teos, 0.1)
        return ctc, ce

    def barrier():
        if world > 1:
            torch.distributed.barrier()

    # initialisation, before the W warmup steps of the contract: the first steps of a process grow the caching allocator's pools,
    # load code objects, create the side streams / event pool and (N > 1) the RCCL communicators; on a cold box one of them can
    # cost tens of milliseconds, which W = 2..3 warmup steps do not always absorb
    for _ in range(6):
        step()
    torch.cuda.synchronize()
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
    if world > 1:
        t = torch.tensor([dt], device=dev, dtype=torch.float64)
        torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
        dt = float(t.item())

    # ---- live per-kernel timing over a second, identical run of the timed region (events add launch overhead, so the
    # headline value above is measured without them) ----
    if trainer is not None:
        trainer.wgrad_stream = False    # (likewise the weight-gradient GEMMs, which the timed step runs on a second side stream)
        trainer.overlap_ctc = False     # serial order for this pass: per-op durations are then uncontended (the CTC branch otherwise
                                        # runs beside the decoder's kernels on a side stream and both would read slower than they are)
    ops.profile_start()
    for _ in range(args.steps):
        step()
    prof = ops.profile_stop()

    if rank == 0:
        frames = world * CFG["B"] * CFG["T"] * args.steps
        kernels = []
        for name, r in prof.items():
            per_ms = r["ms"] / r["calls"]
            hbm = name.startswith(("add_layernorm", "ctc_loss"))
            ach = (r["work"] / r["calls"]) / (per_ms * 1e-3) / (1e9 if hbm else 1e12)
            kernels.append(dict(name=name, calls_per_step=r["calls"] / args.steps, ms_per_call=round(per_ms, 4),
                                ms_per_step=round(r["ms"] / args.steps, 3), bound="hbm" if hbm else "mfma",
                                achieved=round(ach, 2), unit="GB/s" if hbm else "TFLOP/s"))
        kernels.sort(key=lambda k: -k["ms_per_step"])
        dom = kernels[0]
        peak = PEAK_HBM_GBS if dom["bound"] == "hbm" else PEAK_MFMA_BF16_TFLOPS
        roofline = dict(kernel=dom["name"], bound=dom["bound"], achieved=dom["achieved"], peak=peak, unit=dom["unit"],
                        frac=round(dom["achieved"] / peak, 4), traffic=pmc_traffic(dom["name"]),
                        note="achieved = algorithmic FLOPs (or bytes) per launch / mean launch duration from HIP events "
                             "on the launch stream, in a second pass of the same steps with the trainer's CTC side stream "
                             "switched off (uncontended per-op durations; `value` is measured with it on); traffic = HBM bytes per launch from separate rocprofv3 --pmc FETCH_SIZE / "
                             "WRITE_SIZE passes (profiles/r1/pmc_traffic_train_s1.json; FETCH doubled per the gfx950 correction)")
        ctc_k = [k for k in kernels if k["name"].startswith("ctc_loss_fwd")]
        ctc_b = [k for k in kernels if k["name"].startswith("ctc_loss_bwd")]
        what = ("training step (train mode, dropout %g): forward + joint CTC/CE loss + backward + grad all-reduce + Adam" % args.dropout
                if args.mode == "train" else "eval-mode forward + joint CTC/CE loss")
        result = {
            "metric": "fbank frames/sec (CTC_Transformer d256 h4 enc12/dec6, %s)" % what,
            "value": round(frames / dt, 1), "unit": "frames/s", "n_gpus": world, "steps": args.steps, "warmup": args.warmup,
            "ms_per_step": round(dt / args.steps * 1e3, 3), "higher_is_better": True, "scaling": "weak", "vs_baseline": None,
            "dtype": args.precision, "data": "synthetic",
            "config": {"workload": "S1: CTC_Transformer d_model=256 h=4 d_inner=2048 enc12/dec6 V=4234, per-GPU B=32 x T=1000 x 80 "
                                   "fbank, U=50, " + what, "global_batch": world * CFG["B"],
                       "seq_len": CFG["T"], "parallelism": "dp%d" % world},
            "roofline": roofline,
            "ctc": ({"ms_per_step_fwd": ctc_k[0]["ms_per_step"], "fwd_GBps": ctc_k[0]["achieved"],
                     "fwd_frac_of_hbm_peak": round(ctc_k[0]["achieved"] / PEAK_HBM_GBS, 4),
                     "ms_per_step_bwd": (ctc_b[0]["ms_per_step"] if ctc_b else None),
                     # bytes: the logits read twice (f32) + the gradient written (bf16 image in the trainer)
                     "fwd_bwd_GBps": (round((2 * 4.0 + (2.0 if trainer is not None else 4.0)) * CFG["B"] * CFG["T"] * CFG["vocab_size"] / ((ctc_k[0]["ms_per_call"] + ctc_b[0]["ms_per_call"]) * 1e-3) / 1e9, 1)
                                      if ctc_b else None)} if ctc_k else None),
            "kernels": kernels[:12], "op_ms_total": round(sum(k["ms_per_step"] for k in kernels), 3),
        }
        if world == 1 and not args.no_cpu_baseline:
            cb, ref_ctc, ref_ce, ref_ctc_logits, ref_logits = cpu_baseline(model, x, lens, tg, n_utt=16)   # ~15-20 s of host work
            result["cpu_baseline"] = cb
            # sanity: the GPU result on the same utterance agrees with the oracle (bf16 tolerance); not timed
            # (in train mode the weights have moved since `cb` copied them: cpu_baseline() reads the current weights)
            model.eval()                   # the oracle run above is the eval-mode forward
            with torch.no_grad():
                l1, cl1, (lg1, te1) = model(x[:16], lens[:16], tg[:16])
            import numpy as np
            result["parity_vs_oracle_max_abs"] = {
                "ctc_logits": float(np.abs(cl1.float().cpu().numpy() - ref_ctc_logits).max()),
                "logits": float(np.abs(lg1.float().cpu().numpy() - ref_logits).max())}
        print(json.dumps(result))
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
