"""1-rank RCCL sanity on a GPU box: the trainer's bucketed async all-reduce over views of the flat gradient buffer (backend nccl)."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "29533")
import torch, torch.distributed as dist
dist.init_process_group("nccl", rank=0, world_size=1)
torch.cuda.set_device(0)
import asr_amd, bench
from asr_amd import trainer as T
dev = torch.device("cuda", 0)
model = bench.build_model(asr_amd, dev, 0.1, True)
x, lens, tg = bench.make_batch(dev, 0)
tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
tr.buckets.world = 2          # force the all-reduce launches (sum over 1 rank = identity; Adam then averages by 1/2: only the plumbing is under test)
l0 = [float(v) for v in tr.step(x, lens, tg, max_target_len=50)[:1]]
for _ in range(3): out = tr.step(x, lens, tg, max_target_len=50)
torch.cuda.synchronize()
print("rccl sanity ok: launch order", tr.buckets.launch_order, "loss", l0, [float(v) for v in out[:1]])
dist.destroy_process_group()
if "--graph" in sys.argv:      # are RCCL collectives capturable in a hipGraph on this stack?  (1 rank, all-reduce forced, whole step captured)
    dist.init_process_group("nccl", rank=0, world_size=1)
    model = bench.build_model(asr_amd, dev, 0.1, True)
    tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1)
    tr.buckets.world = 2
    for _ in range(6): out = tr.step_graphed(x, lens, tg, max_target_len=50)
    torch.cuda.synchronize()
    print("rccl in graph:", "captured" if tr.graph_active() else ("NOT captured: %s" % tr._graph_failed), "loss", [float(v) for v in out[:1]])
    dist.destroy_process_group()
if "--exec" in sys.argv:
    # The all-reduce inside the C launch loop (csrc/collective.hip + graph_exec.hip), as far as one GPU can show it: a 1-rank RCCL
    # communicator owned by libasr_hip.so, every gradient bucket a collective node of the captured step.  Replay time and the host's
    # queueing time per step, against the same step without the nodes.
    import time
    from asr_amd import ops
    res = {}
    for force in (False, True):
        torch.manual_seed(0)
        model = bench.build_model(asr_amd, dev, 0.1, True)
        tr = asr_amd.Trainer(model, k=0.2, warmup_steps=4000, label_smoothing=0.1, force_collective=force)
        for _ in range(6): out = tr.step_graphed(x, lens, tg, max_target_len=50)
        torch.cuda.synchronize()
        assert tr.graph_active() and tr._graphx is not None, tr._graph_failed
        best = None
        for rot in range(3):
            tr._graphx.set_rotation(rot)
            for _ in range(3): tr.step_graphed(x, lens, tg, max_target_len=50)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); host = 0.0
            for _ in range(20):
                h0 = time.perf_counter()
                out = tr.step_graphed(x, lens, tg, max_target_len=50)
                host += time.perf_counter() - h0
            torch.cuda.synchronize()
            ms = (time.perf_counter() - t0) / 20 * 1e3
            if best is None or ms < best[0]:
                best = (ms, host / 20 * 1e6, rot)
        res[force] = dict(ms_per_step=round(best[0], 3), host_us_per_step=round(best[1], 1), rotation=best[2], info=tr._graphx.info,
                          loss=[round(float(v), 4) for v in out])
        del tr, model
    ver = __import__("ctypes").c_int()
    ops.lib().asr_rccl_version(__import__("ctypes").byref(ver))
    print("executor all-reduce:", {"rccl_version": ver.value, "without": res[False], "with_collective_nodes": res[True],
                                   "ratio": round(res[True]["ms_per_step"] / res[False]["ms_per_step"], 4)})
