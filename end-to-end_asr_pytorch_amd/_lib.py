"""ctypes binding of libasr_hip.so (the C-ABI declared in include/asr_hip.h).

There is deliberately NO fallback: if the shared library is missing or a call fails, the op raises.
`build_library()` compiles the HIP sources in-tree with hipcc for gfx950 (cross-compiles without a GPU).
"""
import ctypes
import os
import subprocess

_HERE = os.path.dirname(os.path.abspath(__file__))
CSRC = os.path.join(_HERE, "csrc")
LIB_PATH = os.environ.get("ASR_AMD_LIB") or os.path.join(CSRC, "libasr_hip.so")     # (ASR_AMD_LIB: another build of the same ABI, for A/B runs)
SOURCES = ["common.hip", "gemm.hip", "ffn.hip", "ffn2.hip", "vocab.hip", "graph_exec.hip", "collective.hip", "dgrad_rows.hip", "attention.hip", "attention_fwd4.hip", "attention_bwd.hip", "attention_bwd4.hip", "norm_embed.hip", "conv.hip", "ctc.hip", "ce.hip", "cif.hip",
           "backward.hip", "wgrad.hip", "fused_small.hip", "cif_train.hip", "decode.hip", "decode_blocks.hip", "input.hip"]
EXTRA_FLAGS = {"cif.hip": ["-ffp-contract=off"],  # bit-exact CIF: product and sum rounded separately, like the reference
               # ffn2.hip: the phase in front of the generated loop keeps its accumulators in VGPRs (the loop's block owns a0-a63 and the
               # compiler places its own AGPRs behind them: 176 + 64 + 64 would not fit a wave at two per SIMD)
               "ffn2.hip": ["-mllvm", "-amdgpu-mfma-vgpr-form=1"]}
HEADER = os.path.join(os.path.dirname(_HERE), "include", "asr_hip.h")

_vp, _i, _i64, _f, _u = ctypes.c_void_p, ctypes.c_int, ctypes.c_int64, ctypes.c_float, ctypes.c_uint


class Dropout(ctypes.Structure):
    """asr_dropout_t (include/asr_hip.h), passed by value.  thr16 == 0 disables dropout."""
    _fields_ = [("thr16", ctypes.c_uint32), ("key0", ctypes.c_uint32), ("key1", ctypes.c_uint32), ("salt", ctypes.c_void_p)]


NO_DROP = Dropout(0, 0, 0, None)
_dr = Dropout

# name -> argtypes (restype is int unless noted).  Kept in lock-step with include/asr_hip.h; tests/test_abi.py checks
# every prototype in the header is exported by the .so and listed here.
SIGNATURES = {
    "asr_gemm_nt": [_vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _vp, _i, _i64, _i, _i, _i, _u],
    "asr_gemm_nt_ex": [_vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _vp, _i, _i64, _i, _i, _i, _u, _vp, _i64, _vp, _i64, _vp, _i64],
    "asr_gemm_nn": [_vp, _vp, _i, _i64, _vp, _i64, _vp, _vp, _i, _i64, _i, _i, _i, _vp, _i64, _vp, _i64, _i],
    "asr_attention_bwd_dq": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _f, _dr, _vp],
    "asr_attention_bwd_dkv": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _dr, _vp],
    "asr_attention_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _f, _dr, _vp],
    "asr_attention_bwd_workspace_floats": [_i, _i, _i],
    "asr_attention_dropmask": [_vp, _dr, _i, _i, _i, _i, _vp],
    "asr_attention_dropmask_multi": [_vp, _i, _vp, _vp, _i, _i, _i, _i],
    "asr_gemm_add_layernorm_small": [_vp, _vp, _i64, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _dr],
    "asr_attention_dropmask_words": [_i, _i, _i, _i],
    "asr_graphx_create": [_vp, _i, _vp],
    "asr_graphx_launch": [_vp, _vp],
    "asr_graphx_set_rotation": [_vp, _i],
    "asr_graphx_place_streams": [_vp, _vp, _i],
    "asr_graphx_info": [_vp, _vp, _vp, _vp, _vp],
    "asr_graphx_destroy": [_vp],
    "asr_rccl_load": [ctypes.c_char_p],
    "asr_rccl_version": [_vp],
    "asr_rccl_unique_id": [_vp],
    "asr_rccl_comm_create": [_vp, _i, _i, _vp],
    "asr_rccl_comm_destroy": [_vp],
    "asr_rccl_comm_abort": [_vp],
    "asr_rccl_all_reduce_f32": [_vp, _vp, ctypes.c_longlong, _vp],
    "asr_rccl_comm_check": [_vp],
    "asr_collective_mark": [_vp, ctypes.c_longlong, _i, _vp],
    "asr_graphx_set_collective": [_vp, _vp, _vp, _vp],
    "asr_graphx_collectives": [_vp, _vp, _vp],
    "asr_ffn_bits_words": [_i, _i],
    "asr_ffn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _dr],
    "asr_attn_ffn_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _f, _dr, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp,
                         _vp, _vp, _i, _i, _i, _i, _f, _dr],
    "asr_proj_ln_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _dr],
    "asr_ffn_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asr_dgrad_rows": [_vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i],
    "asr_dgrad_rows_ln": [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dr],
    "asr_add_layernorm_bwd_y": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _dr],
    "asr_ffn_bwd_ln": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _dr],
    "asr_add_layernorm_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _dr, _dr],
    "asr_gemm_tn": [_vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _i],
    "asr_debug_poison_lds": [_vp, _vp],
    "asr_set_deterministic": [_i],
    "asr_launch_budget": [_i],
    "asr_streams_share_queue": [_vp, _vp, _vp],
    "asr_event_create": [_vp],
    "asr_stream_order_after": [_vp, _vp, _vp],
    "asr_event_destroy": [_vp],
    "asr_timer_create": [_vp],
    "asr_timer_record": [_vp, _vp],
    "asr_timer_elapsed_ms": [_vp, _vp, _vp],
    "asr_gemm_tn_ws_bytes": [_i, _i, _i, _i],
    "asr_gemm_tn_ws_group": [_vp, _i, _vp, _i],
    "asr_gemm_tn_ws_group_wgs": [_vp, _i, _vp, _i, _i],
    "asr_gemm_tn_ws": [_vp, _vp, _i64, _vp, _i64, _vp, _i64, _i, _i, _i, _i, _vp, _i, _vp, _i64, _i],
    "asr_colsum": [_vp, _vp, _i, _i64, _i, _i, _vp, _i],
    "asr_embed_bwd": [_vp, _vp, _vp, _i, _i, _i, _i, _vp, _dr],
    "asr_dropout_apply": [_vp, _vp, _vp, _i, _i, _i, _dr],
    "asr_adam_step": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _f, _f, _f, _f, _i, _f],
    "asr_step_tick": [_vp, _vp, _f, _f, _f, _f, _f],
    "asr_adam_step_dev": [_vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _f, _f, _f, _f],
    "asr_proj_heads": [_vp, _vp, _i, _i64, _vp, _i, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _i, _f],
    "asr_attention_fwd": [_vp, _vp, _vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _vp, _i, _dr, _vp],
    "asr_add_layernorm_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _dr, _dr],
    "asr_embed_pe_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _dr],
    "asr_decoder_targets": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i64, _i64],
    "asr_decoder_cif_targets": [_vp, _vp, _vp, _vp, _i, _i, _i64],
    "asr_conv_sub0_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asr_conv_sub1_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asr_conv_im2col": [_vp, _vp, _i, _i, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asr_conv_col2im_relu": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i],
    "asr_conv_col2im_relu_f32": [_vp, _vp, _i, _vp, _vp, _i, _i, _i, _i, _i],
    "asr_conv_sub1_bwd_x": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asr_conv_sub1_bwd_w": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asr_conv_sub1_bwd_w_workspace_floats": [],
    "asr_conv_sub0_bwd_w": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i],
    "asr_ctc_workspace_stride": [_i],
    "asr_ctc_loss_fwd": [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i],
    "asr_ctc_loss_mean_fwd": [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _i, _vp],
    "asr_ctc_counter_words": [_i, _i, _i],
    "asr_ctc_mean": [_vp, _vp, _vp, _i, _vp],
    "asr_ctc_loss_bwd": [_vp, _vp, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp],
    "asr_ctc_loss_bwd_ex": [_vp, _vp, _i, _i64, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i64, _vp],
    "asr_vocab_proj_ctc": [_vp, _vp, _vp, _vp, _i64, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asr_ctc_loss_fwd_table": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "asr_ce_loss_fwd": [_vp, _vp, _i64, _vp, _i, _i, _f, _vp, _vp],
    "asr_ce_mean": [_vp, _vp, _vp, _i, _vp],
    "asr_ce_mean_masked": [_vp, _vp, _vp, _vp, _i, _vp],
    "asr_token_mask": [_vp, _vp, _vp, _i, _i, _f, _i, _vp, _vp],
    "asr_ce_loss_bwd": [_vp, _vp, _i64, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _i, _i64],
    "asr_cif_scan_fwd": [_vp, _vp, _i, _i, _f, _vp, _vp, _vp, _vp, _vp, _vp],
    "asr_cif_gather_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp, _vp, _vp],
    "asr_cif_scan_bwd": [_vp, _vp, _vp, _vp, _i, _i, _vp],
    "asr_cif_gather_fwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _vp],
    "asr_assigner_tail_fwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp],
    "asr_lfr_stack": [_vp, _vp, _vp, _i, _i, _i, _i, _i, _vp, _vp],
    "asr_spec_aug": [_vp, _vp, _vp, _i, _i, _i, _vp, _i, _i, _i, _i, _vp, _vp],
    "asr_argmax_rows": [_vp, _vp, _i64, _i, _i, _vp],
    "asr_log_softmax_rows": [_vp, _vp, _i64, _i, _i, _vp, _i64],
    "asr_ctc_greedy_reduce": [_vp, _vp, _vp, _i, _i, _i, _vp, _vp],
    "asr_attention_bwd_f32": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i64, _vp, _vp, _i64, _i, _i, _i, _i, _vp, _i, _f],
    "asr_topk_rows": [_vp, _vp, _i64, _i, _i, _i, _vp, _vp],
    "asr_lsm_topk_rows": [_vp, _vp, _i64, _i, _i, _i, _i, _vp, _vp],
    "asr_beam_prune": [_vp, _vp, _vp, _vp, _i, _i, _vp, _vp, _vp],
    "asr_decode_embed": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i],
    "asr_kv_cache_put": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asr_decode_advance": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asr_decode_block_workspace_bytes": [_i],
    "asr_decode_ffn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _f, _vp, _vp, _vp, _vp, _f],
    "asr_decode_self_attn": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _f, _vp, _vp, _vp, _i, _f],
    "asr_beam_cat_frames": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i, _i],
    "asr_beam_step": [_vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _i],
    "asr_beam_reorder_cache": [_vp, _vp, _vp, _vp, _i, _i, _i, _i, _i, _i],
    "asr_beam_advance": [_vp, _vp, _vp, _i, _vp, _i, _vp, _vp],
    "asr_add2d": [_vp, _vp, _i64, _vp, _i64, _i, _i],
    "asr_add_transposed": [_vp, _vp, _vp, _i, _i, _i, _i64],
    "asr_relu_mask_mul": [_vp, _vp, _vp, _i, _vp, _i64],
    "asr_conv1d_overlap_add": [_vp, _vp, _i, _i, _i, _vp],
    "asr_assigner_tail_bwd": [_vp, _vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp],
    "asr_cif_rescale_fwd": [_vp, _vp, _vp, _vp, _i, _i, _i, _vp, _vp, _vp, _vp],
    "asr_cif_rescale_bwd": [_vp, _vp, _vp, _vp, _vp, _vp, _i, _i, _vp],
    "asr_cast_f32_bf16": [_vp, _vp, _vp, _i64],
    "asr_mask_rows": [_vp, _vp, _vp, _i, _i, _i, _i64],
}

_lib = None


def hipcc_path():
    for p in (os.environ.get("HIPCC"), "/opt/rocm/bin/hipcc", "hipcc"):
        if p and (os.path.isabs(p) and os.path.exists(p) or not os.path.isabs(p)):
            return p
    return "hipcc"


def existing_sources():
    return [s for s in SOURCES if os.path.exists(os.path.join(CSRC, s))]


def build_library(force=False, verbose=False):
    """hipcc --offload-arch=gfx950 -shared -fPIC csrc/*.hip -o csrc/libasr_hip.so (rebuilt only when stale)."""
    srcs = [os.path.join(CSRC, s) for s in existing_sources()]
    incs = sorted(os.path.join(CSRC, f) for f in os.listdir(CSRC) if f.endswith(".inc"))     # generated instruction streams (tools/gen_attn_*.py)
    deps = srcs + incs + [os.path.join(CSRC, "asr_common.h"), HEADER]
    if not force and os.path.exists(LIB_PATH) and all(os.path.getmtime(LIB_PATH) >= os.path.getmtime(d) for d in deps):
        return LIB_PATH
    objs = []
    procs = []
    os.makedirs(os.path.join(CSRC, "build"), exist_ok=True)
    for s in srcs:
        o = os.path.join(CSRC, "build", os.path.basename(s) + ".o")
        objs.append(o)
        if not force and os.path.exists(o) and all(
                os.path.getmtime(o) >= os.path.getmtime(d) for d in [s, deps[-2], deps[-1]] + incs):
            continue
        cmd = [hipcc_path(), "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC"] + EXTRA_FLAGS.get(os.path.basename(s), []) + \
            os.environ.get("ASR_AMD_EXTRA_HIPCC_FLAGS", "").split() + ["-c", s, "-o", o]      # (diagnostic builds: -DHEADS_ABLATE, -DFFN_STAMP ...)
        if verbose:
            print(" ".join(cmd))
        procs.append((cmd, subprocess.Popen(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)))
    for cmd, p in procs:
        out, _ = p.communicate()
        if p.returncode != 0:
            raise RuntimeError("hipcc failed: %s\n%s" % (" ".join(cmd), out.decode(errors="replace")))
    cmd = [hipcc_path(), "--offload-arch=gfx950", "-shared", "-fPIC", "-o", LIB_PATH] + objs
    r = subprocess.run(cmd, stdout=subprocess.PIPE, stderr=subprocess.STDOUT)
    if r.returncode != 0:
        raise RuntimeError("link failed: %s\n%s" % (" ".join(cmd), r.stdout.decode(errors="replace")))
    return LIB_PATH


def lib():
    """Load libasr_hip.so; raise loudly when it is absent (no CPU / eager fallback exists by design)."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise RuntimeError(
                "libasr_hip.so not found at %s - run `python -c 'import __graft_entry__ as g; g.build()'` "
                "(the MI355X path has no fallback)" % LIB_PATH)
        L = ctypes.CDLL(LIB_PATH)
        for name, argtypes in SIGNATURES.items():
            fn = getattr(L, name)
            fn.argtypes = argtypes
            fn.restype = ctypes.c_int
        L.asr_attention_dropmask_words.restype = ctypes.c_int64
        L.asr_attention_bwd_workspace_floats.restype = ctypes.c_int64
        L.asr_ffn_bits_words.restype = ctypes.c_int64
        L.asr_gemm_tn_ws_bytes.restype = ctypes.c_int64
        L.asr_ctc_counter_words.restype = ctypes.c_int64
        L.asr_conv_sub1_bwd_w_workspace_floats.restype = ctypes.c_int64
        L.asr_decode_block_workspace_bytes.restype = ctypes.c_int64
        L.asr_last_error.restype = ctypes.c_char_p
        L.asr_version.restype = ctypes.c_int
        _lib = L
    return _lib


def check(rc, what):
    if rc != 0:
        msg = lib().asr_last_error().decode(errors="replace")
        raise RuntimeError("%s failed (rc=%d): %s" % (what, rc, msg))
