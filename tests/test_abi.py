"""The C-ABI library builds for gfx950, loads without a GPU, and exports exactly what include/asr_hip.h declares."""
import os
import re
import subprocess

import pytest

import asr_amd
from asr_amd import _lib

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def header_prototypes():
    src = open(os.path.join(ROOT, "include", "asr_hip.h")).read()
    src = re.sub(r"/\*.*?\*/", "", src, flags=re.S)
    protos = {}
    for m in re.finditer(r"\b(int64_t|int|const char\*)\s+(asr_\w+)\s*\(([^;]*?)\)\s*;", src, flags=re.S):
        args = [a.strip() for a in m.group(3).split(",")] if m.group(3).strip() != "void" else []
        protos[m.group(2)] = args
    return protos


def test_library_builds_and_loads():
    path = asr_amd.build_library()
    assert os.path.exists(path)
    L = asr_amd.lib()
    assert L.asr_version() >= 102
    assert L.asr_last_error() is not None


def test_every_declared_symbol_is_exported_and_bound():
    protos = header_prototypes()
    assert len(protos) >= 20
    out = subprocess.run(["nm", "-D", "--defined-only", _lib.LIB_PATH], capture_output=True, text=True, check=True).stdout
    exported = {line.split()[-1] for line in out.splitlines() if " T " in line}
    for name, args in protos.items():
        assert name in exported, "%s declared in asr_hip.h but not exported" % name
        if name in ("asr_version", "asr_last_error"):
            continue
        assert name in _lib.SIGNATURES, "%s has no ctypes signature" % name
        assert len(_lib.SIGNATURES[name]) == len(args), "%s: header has %d args, binding %d" % (
            name, len(args), len(_lib.SIGNATURES[name]))
    extra = {e for e in exported if e.startswith("asr_")} - set(protos) - {"asr_set_error"}
    assert not extra, "exported but undeclared: %s" % extra


def test_invalid_arguments_fail_loudly_without_gpu():
    import ctypes
    L = asr_amd.lib()
    rc = L.asr_gemm_nt(None, None, 0, 8, None, 0, 8, None, None, 0, 8, 4, 4, 8, 0)
    assert rc == -1 and b"gemm" in L.asr_last_error()
    rc = L.asr_ctc_loss_fwd(None, None, 0, None, None, 1, 1, 4, 1, 3, None, None, None, None, None, None, 1)
    assert rc == -1


def test_cpu_tensors_are_rejected_not_silently_computed():
    import pytest
    import torch
    with pytest.raises(RuntimeError, match="no CPU fallback"):
        asr_amd.ops.gemm_nt(torch.zeros(8, 8), torch.zeros(8, 8))


def test_host_side_size_queries_need_no_gpu():
    """The workspace / image size entry points and the mode switch are pure host code: callable on the build box."""
    L = asr_amd.lib()
    tile_f = 128 * 128 + 128
    # FFN-shaped weight gradient at S1: 32 output tiles x 8 M-splits of one slab tile each (csrc/wgrad.hip: tn_v2_plan)
    assert L.asr_gemm_tn_ws_bytes(32000, 256, 2048, 256) == 16 + 8 * 32 * tile_f * 4
    assert L.asr_gemm_tn_ws_bytes(32000, 256, 2048, 0) == L.asr_gemm_tn_ws_bytes(32000, 256, 2048, 256)
    # the decoder's 1632 rows: at least 512 rows per split
    assert L.asr_gemm_tn_ws_bytes(1632, 256, 256, 0) == 16 + 4 * 4 * tile_f * 4
    assert L.asr_gemm_tn_ws_bytes(1632, 256, 200, 0) == 0              # K % 128: not a shape the slab kernel takes
    assert L.asr_ffn_bits_words(32000, 2048) == 32 * 2 * 32000
    assert L.asr_ffn_bits_words(130, 64) == 1 * 2 * 256                # rows padded to the 128-token block
    old = L.asr_set_deterministic(1)
    assert L.asr_set_deterministic(old) == 1


def test_launch_budget_is_per_host_thread_and_restored_by_the_context():
    """asr_launch_budget (asr_hip.h): thread-local state, the previous value comes back; ops.launch_budget restores it on exit, also on an
    exception, and nests."""
    import threading
    from asr_amd import ops
    L = _lib.lib()
    assert L.asr_launch_budget(0) == 0
    with ops.launch_budget(96):
        assert L.asr_launch_budget(96) == 96 and ops._BUDGET.cus == 96
        with ops.launch_budget(0):                    # "no budget" inside a budgeted region
            assert L.asr_launch_budget(0) == 0 and ops._BUDGET.cus == 0
        assert L.asr_launch_budget(96) == 96 and ops._BUDGET.cus == 96
        seen = []
        t = threading.Thread(target=lambda: seen.append((L.asr_launch_budget(0), ops._BUDGET.cus)))      # another host thread starts with none, on both halves
        t.start()
        t.join()
        assert seen == [(0, 0)]
    assert L.asr_launch_budget(0) == 0 and ops._BUDGET.cus == 0
    try:
        with ops.launch_budget(64):
            raise RuntimeError("x")
    except RuntimeError:
        pass
    assert L.asr_launch_budget(0) == 0 and ops._BUDGET.cus == 0
    assert L.asr_launch_budget(-5) == 0 and L.asr_launch_budget(0) == 0      # negative = none


def test_rccl_comm_abort_drops_the_handle_and_evicts_the_cache_entry():
    """a communicator the graph executor borrowed is aborted by its OWNER (ops.RcclComm.abort) - the handle is dropped and the
    per-group cache forgets it, so neither the atexit teardown nor a later rccl_comm() touches freed memory.  (No RCCL call is made here:
    a null-library abort is a no-op in C; what is checked is the host-side ownership logic.)"""
    from asr_amd import ops
    c = ops.RcclComm(None, 1, 0)
    class _Key:
        pass
    key = _Key()
    ops._RCCL["comms"][key] = c
    c.handle = 0          # "no communicator": abort must still evict
    c.abort()
    assert key not in ops._RCCL["comms"] and all(v is not c for v in ops._RCCL["comms"].values())
    assert c.handle is None
    with pytest.raises(RuntimeError, match="aborted or destroyed"):
        c.check()
    with pytest.raises(RuntimeError, match="aborted or destroyed"):
        c._live()
    assert ops.ERR_COLLECTIVE_STEP == -6
    hdr = open(_lib.HEADER).read()
    assert "#define ASR_ERR_COLLECTIVE_STEP (-6)" in hdr
