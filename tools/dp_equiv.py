#!/usr/bin/env python3
"""Data-parallel equivalence: N ranks x B/N utterances produce the gradient of 1 rank x B utterances.

Run under torch.distributed.run with N ranks (backend from ASR_AMD_DIST_BACKEND, default nccl; ASR_AMD_DEVICE pins every rank to
one GPU for the 1-GPU test rig).  Every rank builds the same seeded model twice: once under a Trainer on a single-rank group fed
the WHOLE batch (the reference point: src/transformer/solver.py:83-93 on the concatenated batch) and once under the data-parallel
Trainer fed its shard.  Target lengths differ across shards, so the CE denominator n_word (loss.py:22-25) differs per rank - the
case where DDP's mean of per-rank means is not the global mean.  Prints one JSON line per rank with the worst per-parameter
relative L2 deviation for exact_global_mean on / off."""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import torch.distributed as dist  # noqa: E402


def main():
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    dist.init_process_group(os.environ.get("ASR_AMD_DIST_BACKEND", "nccl"))
    dev_index = int(os.environ.get("ASR_AMD_DEVICE", os.environ.get("LOCAL_RANK", "0")))
    torch.cuda.set_device(dev_index)
    dev = torch.device("cuda", dev_index)
    import asr_amd
    solo = None
    for r in range(world):                      # new_group is collective: every rank creates every single-rank group
        g = dist.new_group([r])
        if r == rank:
            solo = g

    V, B, T, U = 50, 4 * world, 96, 9
    gen = torch.Generator().manual_seed(7)
    x = torch.randn(B, T, 80, generator=gen)
    lens = torch.randint(T // 2, T + 1, (B,), generator=gen)
    lens[::2] = T                                # every shard's longest utterance fills the padded length (utils.py:126-127)
    tg = torch.randint(4, V - 1, (B, U), generator=gen)
    ul = torch.tensor([U if (b // (B // world)) % 2 == 0 else 2 + b % 3 for b in range(B)])   # long targets on even ranks, short on odd
    ul[0] = U
    tg = tg * (torch.arange(U)[None, :] < ul[:, None])
    x, lens, tg = x.to(dev), lens.to(dev), tg.to(dev)

    out = {}

    def build():
        torch.manual_seed(0)
        m = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0),
                                    asr_amd.Decoder(2, 3, V, 2, 2, 64, 128, dropout=0.0)).to(dev)
        return m.train()

    def grads(trainer, xs, ls, ts, umax):
        trainer._fwd_bwd(xs, ls, ts, None, umax)
        torch.cuda.synchronize()
        return {n: p.grad.detach().float().clone() / trainer.world for n, p in trainer.model.named_parameters()}

    def grads_graphed(trainer, xs, ls, ts, umax):
        """--graph: the same gradient out of a CAPTURED step replayed by the multi-stream executor, whose C loop makes the all-reduce
        calls at the buckets' marker nodes (RCCL; the gloo rig hands it torch.distributed's all-reduce as a callback).  The flat
        gradient still holds the summed gradient after the step: Adam reads it, the next step zeroes it."""
        trainer._eager_steps = 2                 # (skip the two eager warm-up steps: one step on both sides)
        trainer.step_graphed(xs, ls, ts, max_target_len=umax)
        torch.cuda.synchronize()
        assert trainer.graph_active() and trainer._graphx is not None, trainer._graph_failed
        assert trainer._graphx.info["collectives"] >= len(trainer.buckets.ranges), trainer._graphx.info
        out["executor"] = trainer._graphx.info
        res = {n: p.grad.detach().float().clone() / trainer.world for n, p in trainer.model.named_parameters()}
        # a rank whose batch signature changes after the capture must fail loudly, not re-capture (its peers would be waiting in the
        # executor's all-reduces on another communicator - ADVICE r4); raised before anything is queued, on every rank alike
        try:
            trainer.step_graphed(xs, ls, ts, max_target_len=umax + 1)
            raise AssertionError("a changed batch signature re-captured the data-parallel step silently")
        except RuntimeError as e:
            assert "batch signature changed" in str(e), e
            out["recapture_refused"] = True
        return res

    graphed = "--graph" in sys.argv
    if graphed:
        asr_amd.set_precision("bf16")
    ref = grads(asr_amd.Trainer(build(), process_group=solo), x, lens, tg, U)
    sl = slice(rank * (B // world), (rank + 1) * (B // world))
    out.update({"rank": rank, "world": world})
    for exact in (True, False):
        tr = asr_amd.Trainer(build(), exact_global_mean=exact)
        assert tr.world == world
        got = (grads_graphed if graphed else grads)(tr, x[sl].contiguous(), lens[sl].contiguous(), tg[sl].contiguous(), int(ul[sl].max()))
        worst, name = 0.0, ""
        for n, g in got.items():
            d = float((g - ref[n]).norm() / (ref[n].norm() + 1e-12))
            if d > worst:
                worst, name = d, n
        out["exact" if exact else "ddp"] = {"worst_rel_l2": worst, "param": name}
    print(json.dumps(out), flush=True)
    dist.destroy_process_group()


if __name__ == "__main__":
    main()
