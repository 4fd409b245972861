"""The N>1 path on CPU: two gloo ranks run the bucketed, overlapped gradient all-reduce (trainer.GradBuckets) over a flat
buffer with the same parameter ordering the GPU trainer uses, and end up with identical, correctly summed gradients."""
import os
import socket

import numpy as np
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

import asr_amd
from asr_amd.trainer import GradBuckets, _param_order, flat_offsets


def _free_port():
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.manual_seed(0)
    model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
    params = _param_order(model)
    offs, n = flat_offsets(params)
    flat = torch.zeros((n + 63) // 64 * 64)
    for p, off in zip(params, offs):
        p.grad = flat[off:off + p.numel()].view(p.shape)
    buckets = GradBuckets(flat, params, offs, n, n_buckets=4)
    assert buckets.world == world
    assert buckets.ranges[0][0] == 0 and buckets.ranges[-1][1] == flat.numel()
    assert all(buckets.ranges[i][1] == buckets.ranges[i + 1][0] for i in range(len(buckets.ranges) - 1))
    # "backward": parameters become final in reverse forward order, one module's worth at a time
    g = torch.Generator().manual_seed(100 + rank)
    buckets.start()
    for p in reversed(params):
        p.grad.copy_(torch.randn(p.shape, generator=g))
        buckets.on_done((p,))
    buckets.finish()
    # every bucket launched exactly once, last bucket (decoder side) first
    assert sorted(buckets.launch_order) == list(range(len(buckets.ranges)))
    assert buckets.launch_order[0] == len(buckets.ranges) - 1
    out[rank] = flat.clone()
    dist.destroy_process_group()


def test_two_rank_bucketed_allreduce_gloo():
    world = 2
    port = _free_port()
    mgr = mp.Manager()
    out = mgr.dict()
    mp.spawn(_worker, args=(world, port, out), nprocs=world, join=True)
    a, b = out[0], out[1]
    np.testing.assert_array_equal(a.numpy(), b.numpy())        # both ranks hold the same reduced gradient
    # and it is the sum of what each rank produced
    exp = None
    for rank in range(world):
        g = torch.Generator().manual_seed(100 + rank)
        torch.manual_seed(0)
        model = asr_amd.CTC_Transformer(asr_amd.Encoder(80, 2, 2, 64, 128, dropout=0.0), asr_amd.Decoder(2, 3, 50, 2, 2, 64, 128, dropout=0.0))
        params = _param_order(model)
        chunks = {}
        for p in reversed(params):
            chunks[id(p)] = torch.randn(p.shape, generator=g).reshape(-1)
        flat = torch.cat([chunks[id(p)] for p in params])
        exp = flat if exp is None else exp + flat
    np.testing.assert_allclose(a[:exp.numel()].numpy(), exp.numpy(), rtol=1e-6, atol=1e-6)


def test_never_ready_bucket_is_an_error():
    p = [torch.nn.Parameter(torch.zeros(8)), torch.nn.Parameter(torch.zeros(8))]
    flat = torch.zeros(64)
    b = GradBuckets(flat, p, [0, 8], 16, n_buckets=2)
    b.start()
    b.on_done((p[0],))
    try:
        b.finish()
        assert False, "expected an error"
    except RuntimeError as e:
        assert "never became ready" in str(e)


def _settle_worker(rank, world, port, out):
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import bench
    # rank 0 would settle after two groups on its own timings, rank 1 after four: the collective decision makes both run four
    mine = {0: [10.0, 10.05, 10.0, 10.0, 10.0], 1: [12.0, 11.0, 10.5, 10.45, 10.4]}[rank]
    calls = []

    def group():
        calls.append(1)
        dist.all_reduce(torch.zeros(1))          # (every step of the real loop holds collectives: an uneven count would hang right here)
        return mine[len(calls) - 1]
    seen = bench.settle_groups(group, world, torch.device("cpu"))
    out[rank] = (len(calls), seen)
    dist.destroy_process_group()


def test_bench_settle_loop_runs_the_same_number_of_groups_on_every_rank():
    """bench.py's warm-up settles on the MAX over ranks of each group's time (round 4: the per-rank decision was the multi-rank hang)."""
    world = 2
    with mp.Manager() as mgr:
        out = mgr.dict()
        mp.spawn(_settle_worker, args=(world, _free_port(), out), nprocs=world, join=True)
        res = dict(out)
    assert res[0][0] == res[1][0] == 4
    assert res[0][1] == res[1][1] == [12.0, 11.0, 10.5, 10.45]
