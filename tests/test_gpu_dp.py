"""The data-parallel training step with world_size 2 on the GPU box: two ranks share cuda:0 over gloo (the box has one GPU; the
product backend is nccl = RCCL) and run bench.py's real step - bucketed async all-reduce overlapped with the HIP backward, fused
Adam - after which bench.py itself asserts that both ranks hold bit-identical parameters."""
import json
import os
import subprocess
import sys

import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_two_rank_training_step_on_one_gpu():
    env = dict(os.environ, ASR_AMD_DIST_BACKEND="gloo", ASR_AMD_DEVICE="0", HSA_ENABLE_IPC_MODE_LEGACY="0")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", "--nproc-per-node", "2", "--master-addr", "127.0.0.1",
           "--master-port", "29541", os.path.join(ROOT, "bench.py"), "--gpus", "2", "--steps", "2", "--warmup", "1", "--no-cpu-baseline"]
    r = subprocess.run(cmd, env=env, cwd=ROOT, capture_output=True, text=True, timeout=900)
    assert r.returncode == 0, r.stderr[-3000:]
    line = [l for l in r.stdout.splitlines() if l.startswith("{")][-1]
    out = json.loads(line)
    assert out["n_gpus"] == 2 and out["config"]["global_batch"] == 64 and out["value"] > 0
