"""The data-parallel training step on the GPU box.

* world_size 2 sharing cuda:0 over gloo (the development box has one GPU; the product backend is nccl = RCCL): bench.py's real
  step - bucketed async all-reduce overlapped with the HIP backward, fused Adam - after which bench.py itself asserts that both
  ranks hold bit-identical parameters;
* the §8(e) equivalence: 2 ranks x B/2 utterances == 1 rank x B utterances on the concatenated batch (tools/dp_equiv.py), with
  target lengths that differ across ranks so that the CE denominator matters;
* the same two checks over RCCL with one rank per GPU when the node has more than one (skipped on a 1-GPU box)."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def _ngpu():
    import torch
    return torch.cuda.device_count()      # (does not initialise the GPU)


def _run(script_args, nproc, port, gloo_one_gpu):
    import signal
    import time
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY="0", PYTHONFAULTHANDLER="1")
    if gloo_one_gpu:
        # (ASR_AMD_GRAPH_DP=1: the captured data-parallel step is opt-in until it has run on N > 1 GPUs; the rig asks for it so that
        # bench.py's step_auto and dp_equiv.py --graph exercise the executor's collective nodes)
        env.update(ASR_AMD_DIST_BACKEND="gloo", ASR_AMD_DEVICE="0", ASR_AMD_GRAPH_DP=env.get("ASR_AMD_GRAPH_DP", "1"))
    # One attempt, bounded.  (Round 3 retried here: the rig "hung once in a few dozen runs".  tools/dp_hang_hunt.py reproduced it - 6 of
    # 80 runs - and the workers' Python stacks showed one rank in bench.py's barrier and the other still stepping: bench.py ended its
    # settle loop on each rank's OWN timings, so the ranks ran different numbers of steps, i.e. of all-reduces.  The decision is
    # collective now; 0 of 120 runs hang.)  A run that exceeds the limit is aborted with its stacks kept for the post-mortem.
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", str(nproc), "--master-addr", "127.0.0.1",
           "--master-port", str(port)] + script_args
    p = subprocess.Popen(cmd, env=env, cwd=ROOT, stdout=subprocess.PIPE, stderr=subprocess.PIPE, text=True, start_new_session=True)
    try:
        r_stdout, se = p.communicate(timeout=300)
    except subprocess.TimeoutExpired:
        os.killpg(p.pid, signal.SIGABRT)      # faulthandler: every worker prints its Python stacks
        time.sleep(3)
        try:
            os.killpg(p.pid, signal.SIGKILL)
        except ProcessLookupError:
            pass
        _, se = p.communicate()
        try:
            os.makedirs(os.path.join(ROOT, "gpurun_out"), exist_ok=True)
            with open(os.path.join(ROOT, "gpurun_out", "dp_hang_port%d.txt" % port), "w") as f:
                f.write(se)
        except OSError:
            pass
        raise AssertionError("the %d-rank run did not finish in 300 s; stacks:\n%s" % (nproc, se[-6000:]))
    assert p.returncode == 0, se[-3000:]

    class _R:
        stdout = r_stdout
    r = _R()
    out, dec = [], json.JSONDecoder()
    for line in r.stdout.splitlines():          # ranks share the pipe: two records can land on one line
        i = line.find("{")
        while i >= 0:
            try:
                obj, end = dec.raw_decode(line, i)
            except json.JSONDecodeError:
                break
            out.append(obj)
            i = line.find("{", end)
    return out


def test_two_rank_training_step_on_one_gpu():
    out = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 2, 29541, True)[-1]
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0


def _check_equiv(lines, world):
    assert len(lines) == world
    for o in lines:
        # the data-parallel gradient IS the single-process gradient of the concatenated batch (fp32 summation order / bf16
        # rounding noise only) ...
        assert o["exact"]["worst_rel_l2"] < 5e-3, o
        # ... which DDP's mean of per-rank means is not when n_word differs across ranks (the test batch makes it differ by > 2x)
        assert o["ddp"]["worst_rel_l2"] > 5e-2, o


def test_two_ranks_equal_one_rank_on_the_concatenated_batch():
    _check_equiv(_run([os.path.join(ROOT, "tools", "dp_equiv.py")], 2, 29543, True), 2)


def test_two_ranks_through_the_graph_executor_equal_one_rank():
    """The same equivalence with each rank's step CAPTURED and replayed by the multi-stream executor, whose launch loop calls the
    all-reduce at the buckets' collective nodes (csrc/collective.hip; the gloo rig passes torch.distributed's all-reduce as the
    node's callback - RCCL does not take two ranks on one GPU)."""
    lines = _run([os.path.join(ROOT, "tools", "dp_equiv.py"), "--graph"], 2, 29551, True)
    _check_equiv(lines, 2)
    assert all(o["executor"]["collectives"] >= 2 for o in lines), lines
    assert all(o.get("recapture_refused") for o in lines), lines


def test_rccl_communicator_refused_means_every_rank_steps_eagerly(monkeypatch):
    """ASR_AMD_DP_COMM=rccl makes the gloo rig ask for the executor's OWN RCCL communicator: the unique id travels over
    torch.distributed, ncclCommInitRank refuses two ranks on one GPU - on both ranks.  The capture is abandoned, the ranks agree on
    it (an all-reduced flag) and train eagerly; bench.py's parameter checksums still match."""
    monkeypatch.setenv("ASR_AMD_DP_COMM", "rccl")
    out = _run([os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"], 2, 29555, True)[-1]
    assert out["n_gpus"] == 2 and out["value"] > 0
    assert "eager" in out["config"]["launch"] or "queued" in out["config"]["launch"], out["config"]["launch"]


def test_eight_ranks_through_the_graph_executor_equal_one_rank():
    lines = _run([os.path.join(ROOT, "tools", "dp_equiv.py"), "--graph"], 8, 29553, True)
    _check_equiv(lines, 8)


def test_eight_ranks_equal_one_rank_on_the_concatenated_batch():
    """The world size of BASELINE.json configs[4] (8 shards of the utterance batch, 8 gradient buckets each), still over gloo on one GPU."""
    _check_equiv(_run([os.path.join(ROOT, "tools", "dp_equiv.py")], 8, 29549, True), 8)


@pytest.mark.skipif(_ngpu() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
def test_rccl_ranks_equal_one_rank():
    n = min(_ngpu(), 8)
    _check_equiv(_run([os.path.join(ROOT, "tools", "dp_equiv.py")], n, 29545, False), n)


@pytest.mark.skipif(_ngpu() < 2, reason="needs >= 2 GPUs (RCCL over xGMI)")
def test_rccl_training_step_all_gpus():
    n = min(_ngpu(), 8)
    out = _run([os.path.join(ROOT, "bench.py"), "--gpus", str(n), "--steps", "2", "--warmup", "1"], n, 29547, False)[-1]
    assert out["n_gpus"] == n and out["config"]["global_batch"] == 32 * n and out["value"] > 0
