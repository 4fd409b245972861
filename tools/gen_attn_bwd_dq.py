#!/usr/bin/env python3
"""Generator of the attention backward's dQ instruction stream (csrc/attention_bwd_dq_asm.inc, included by attention_bwd4.hip).

The query-stationary half of the flash backward (attention.py:76-84 differentiated), built like tools/gen_attn_bwd.py:

  a wave owns 32 queries (query on the lane); Q and dO fragments stay in AGPRs, dQ^T [64 d x 32 queries] accumulates in AGPRs;
  the workgroup (4 waves = 128 queries) walks the keys in tiles of 64 = two halves A, B of 32 keys:
      S^T  = K.Q^T - lse     (A = K rows from LDS, B = Q fragments; the accumulator starts at the lane's -lse: a persistent register vector)
      dP^T = V.dO^T - delta  (likewise; with dropout it starts at 0, the keep mask and 1 / keep come first)
      dS^T = exp2(S^T) o dP^T, packed to bf16 in place;   dQ^T += K^T.dS^T   (A = transposed reads of the same K tile)
  phase X: the vector port turns half X's (S^T, dP^T) into dS^T while the matrix pipe runs the other half's dQ product (4 MFMAs) and its
  next tile's S^T and dP^T (8 MFMAs).  Also produces the per-row scalars of the dK / dV kernel: -rowsum(dO o O) and -lse, padded
  (asr_attention_bwd_workspace_floats).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_attn_bwd as G
from gen_attn_fwd4 import Stream, v, a, s, rng
from gen_attn_bwd import lds_read, mfma

DROP = False
ABL = 0

SA, SB, DPA, DPB = 0, 16, 32, 48
NL, ND = 64, 80                        # -lse / -delta of the lane's query, broadcast over 16 registers: C operands of the chains' first MFMAs
T = 96
RB, TB = 104, 108                      # LDS read addresses: rows per k-step; transposed [d half][lo / hi]
VOFF = 112                             # LDS-DMA source offsets of this lane's two pieces (K and V tiles alike)
WKA, WKB, SH4, TC, MOFF, QOFF = 114, 115, 116, 117, 118, 119       # (WKA + 8, WKB + 8 = 122, 123: the next tile's keep-bit words)
E0, E1, E2, E3 = 120, 121, 124, 125
DQ, QFR, DOFR, RF, TF = 0, 32, 48, 64, 80      # AGPRs
S_KRS, S_VRS, S_QRS, S_DORS, S_ORS, S_LRS, S_WRS, S_DQRS, S_MKRS = 36, 40, 44, 48, 52, 56, 60, 64, 68
S_T, S_NKT, S_KL, S_LQ, S_H128, S_LDQ, S_KDST, S_SOFF, S_TMP, S_TMP2, S_DSC, S_SCALE, S_STAGE = 72, 73, 74, 75, 76, 77, 78, 79, 80, 81, 82, 83, 84
S_M0, S_M1, S_LQP4, S_MASKT, S_REM, S_SPECIAL, S_ROW0, S_NLOFF = 85, 86, 87, 88, 89, 90, 91, 92
S_RET, S_SPEC_A, S_SPEC_B = 94, 96, 98
SLOT = 16384                           # LDS ring slot: K tile 8192 | V tile 8192
EXPS = [[0, 1], [2, 3], [4, 5], [6, 7], [8], [9], [10], [11], [12], [13], [14], [15]]      # elements whose exp is issued in each of the 12 gaps


def half(X):
    return dict(S=SA, DP=DPA, WK=WKA, H=0, SPEC=S_SPEC_A) if X == "A" else dict(S=SB, DP=DPB, WK=WKB, H=1, SPEC=S_SPEC_B)


def emit_dq_mfma(st, Y, j):
    """dQ^T += K^T . dS^T of half Y: key-step g = j >> 1, d half cb = j & 1; A = transposed K fragment in TF slot j."""
    y = half(Y)
    g, cb = j >> 1, j & 1
    b = y["DP"] + 4 * g
    mfma(st, "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (a(DQ + 16 * cb, 16), a(TF + 4 * j, 4), v(b, 4), a(DQ + 16 * cb, 16)), rng(TF + 4 * j, 4), rng(b, 4))


def emit_sdp(st, Y, j):
    """j = 0..3: S^T k-steps (A = K rows, B = Q fragments); 4..7: dP^T (V rows, dO fragments).  Row fragment in RF slot j & 3."""
    y = half(Y)
    ks, which = j & 3, j >> 2
    acc = y["S"] if which == 0 else y["DP"]
    bfr = (QFR if which == 0 else DOFR) + 4 * ks
    rf = RF + 4 * (j & 3)
    if ks == 0:
        c = v(NL, 16) if which == 0 else ("0" if DROP else v(ND, 16))
    else:
        c = v(acc, 16)
    mfma(st, "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(acc, 16), a(rf, 4), a(bfr, 4), c), rng(rf, 4), [], rng(acc, 16))


def emit_row_read(st, j, Y, slot_off):
    ks, which = j & 3, j >> 2
    rf = RF + 4 * (j & 3)
    off = slot_off + 8192 * which + 4096 * half(Y)["H"]
    lds_read(st, "ds_read_b128 %s, %s offset:%d" % (a(rf, 4), v(RB + ks), off), [RB + ks], rng(rf, 4))


def emit_tr_reads(st, j, Y, slot_off):
    g, cb = j >> 1, j & 1
    tf = TF + 4 * j
    off = slot_off + (2 * half(Y)["H"] + g) * 2048
    lds_read(st, "ds_read_b64_tr_b16 %s, %s offset:%d" % (a(tf, 2), v(TB + 2 * cb), off), [TB + 2 * cb], rng(tf, 2))
    lds_read(st, "ds_read_b64_tr_b16 %s, %s offset:%d" % (a(tf + 2, 2), v(TB + 2 * cb + 1), off), [TB + 2 * cb + 1], rng(tf + 2, 2))


def emit_exp(st, X, i):
    if ABL & 2:
        return
    x = half(X)
    st.valu("v_exp_f32_e32 %s, %s" % (v(T + i % 8), v(x["S"] + i)), [x["S"] + i], [T + i % 8], kind="exp")


def emit_elem(st, X, i, wk):
    if ABL & 2:
        return
    x = half(X)
    ta, dp = T + i % 8, x["DP"] + i
    if DROP:
        st.valu("v_bfe_i32 %s, %s, %d, 1" % (v(TC), v(wk), 8 * (i >> 2) + (i & 3)), [wk], [TC], kind="drop")
        st.valu("v_and_b32_e32 %s, %s, %s" % (v(dp), v(dp), v(TC)), [dp, TC], [dp], kind="drop")              # dropout's mask on dP ...
        st.valu("v_fma_f32 %s, %s, %s, %s" % (v(dp), v(dp), s(S_DSC), v(ND)), [dp, ND], [dp], kind="drop")      # ... / keep - delta
    st.valu("v_mul_f32_e32 %s, %s, %s" % (v(dp), v(ta), v(dp)), [ta, dp], [dp])                                # dS = P (dP - delta)
    if i & 1:
        st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["DP"] + (i >> 1)), v(dp - 1), v(dp)), [dp - 1, dp], [x["DP"] + (i >> 1)], kind="cvt")


def emit_dma_piece(st, p, slot_imm):
    """request p of a tile's four: K pieces 0, 1; V pieces 2, 3."""
    st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_KDST), slot_imm + 8192 * (p >> 1) + (p & 1) * 1024 + 0x10000), kind="salu")


def emit_dma_load(st, p):
    st.raw("buffer_load_dwordx4 %s, %s, %s offen lds" % (v(VOFF + (p & 1)), s(S_KRS if p < 2 else S_VRS, 4), s(S_SOFF)), kind="dma")


def emit_phase(st, X, Y, u, uid):
    x, y = half(X), half(Y)
    st.comment("---- phase %s: dS^T of half %s | dQ product and next (S^T, dP^T) of half %s" % (uid, X, Y))
    if X == "A":
        s_nxt, s_pre = u, u                     # (S, dP) of B(t): tile t; next phase's transposed fragments: A(t)
    else:
        s_nxt, s_pre = (u + 1) & 3, u           # (S, dP) of A(t+1); next phase's transposed fragments: B(t)
    wk = x["WK"] + 8 * (u & 1)
    # the ragged last tile: keys past k_len (their K rows read as zeros) get score -inf, out of line
    st.raw("s_cmp_lg_u32 %s, 0" % s(S_SPECIAL), kind="salu")
    st.raw("s_cbranch_scc0 .Lplain_%s_%s" % (uid, "%="), kind="salu")
    st.raw("s_swappc_b64 %s, %s" % (s(S_RET, 2), s(x["SPEC"], 2)), kind="salu")
    st.label(".Lplain_%s_%s" % (uid, "%="))
    if DROP:
        st.raw("s_waitcnt vmcnt(5)", kind="wait")
        st.raw("v_lshrrev_b32_e32 %s, %s, %s" % (v(wk), v(SH4), v(wk)), kind="drop")
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(x["WK"] + 8 * ((u + 1) & 1)), v(MOFF), s(S_MKRS, 4), s(S_M0 if X == "A" else S_M1)), kind="mload")
    dma_at = {1: 0, 4: 1, 7: 2, 10: 3} if (X == "B" and not (ABL & 8)) else {}
    pending = []
    for gap in range(12):
        if gap < 4:
            emit_dq_mfma(st, Y, gap)
        else:
            emit_sdp(st, Y, gap - 4)
        if gap < 8:
            emit_row_read(st, gap, Y, s_nxt * SLOT)              # row fragments of the (S, dP) MFMAs, four gaps ahead
        else:
            emit_tr_reads(st, gap - 8, X, s_pre * SLOT)          # transposed K fragments of the NEXT phase's dQ product (half X)
        if gap in dma_at:
            emit_dma_piece(st, dma_at[gap], ((u + 3) & 3) * SLOT)
        for i in EXPS[gap][:1]:
            emit_exp(st, X, i)
        if gap in dma_at:
            if ABL & 2:
                st.raw("s_nop 0", kind="nop")
            emit_dma_load(st, dma_at[gap])
        for i in EXPS[gap][1:]:
            emit_exp(st, X, i)
        for i in pending:
            emit_elem(st, X, i, wk)
        pending = EXPS[gap]
    for i in pending:
        emit_elem(st, X, i, wk)
    if dma_at:
        st.raw("s_add_u32 %s, %s, 0x2000" % (s(S_SOFF), s(S_SOFF)), kind="salu")


def emit_special(st, X):
    x = half(X)
    st.label(".Lspecial_%s_%s" % (X, "%="))
    st.raw("s_nop 7")
    st.raw("s_nop 7")
    st.raw("v_sub_u32_e32 %s, %s, %s" % (v(TC), s(S_REM), v(SH4)))            # keys of this tile that exist, minus 4 hh
    st.raw("v_mov_b32_e32 %s, 0xff800000" % v(T + 7))
    for i in range(16):
        key = 32 * x["H"] + (i & 3) + 8 * (i >> 2)
        st.raw("v_cmp_lt_i32_e32 vcc, %d, %s" % (key, v(TC)))
        st.raw("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(x["S"] + i), v(T + 7), v(x["S"] + i)))
    st.raw("s_setpc_b64 %s" % s(S_RET, 2))


def build(drop):
    global DROP
    DROP = drop
    G.DROP = drop
    G.ABL = ABL
    st = Stream()
    U = "%="
    st.comment("==== attention backward dQ (%s): generated by tools/gen_attn_bwd_dq.py - do not edit" % ("train: dropout" if drop else "eval"))
    st.raw("v_and_b32_e32 %s, 0xffff, %%[voff]" % v(VOFF))
    st.raw("v_lshrrev_b32_e32 %s, 16, %%[voff]" % v(VOFF + 1))
    for i, nm in enumerate(("rb01", "rb23")):
        st.raw("v_and_b32_e32 %s, 0xffff, %%[%s]" % (v(RB + 2 * i), nm))
        st.raw("v_lshrrev_b32_e32 %s, 16, %%[%s]" % (v(RB + 2 * i + 1), nm))
    for i, nm in enumerate(("tb01", "tb23")):
        st.raw("v_and_b32_e32 %s, 0xffff, %%[%s]" % (v(TB + 2 * i), nm))
        st.raw("v_lshrrev_b32_e32 %s, 16, %%[%s]" % (v(TB + 2 * i + 1), nm))
    st.raw("v_mov_b32_e32 %s, %%[qoff]" % v(QOFF))
    st.raw("v_mov_b32_e32 %s, %%[tokoff]" % v(E0))
    for name, reg in (("kb", S_KRS), ("vb", S_VRS), ("qb", S_QRS), ("dob", S_DORS), ("ob", S_ORS), ("lb", S_LRS), ("wb", S_WRS), ("dqb", S_DQRS)):
        st.raw("s_mov_b64 %s, %%[%s]" % (s(reg, 2), name))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(reg + 1), s(reg + 1)))
        st.raw("s_mov_b32 %s, 0x00020000" % s(reg + 3))
    for name, reg in (("kl", S_KL), ("lq", S_LQ), ("h128", S_H128), ("ldq2", S_LDQ), ("dsc", S_DSC), ("scale", S_SCALE), ("row0", S_ROW0), ("nloff", S_NLOFF)):
        st.raw("s_mov_b32 %s, %%[%s]" % (s(reg), name))
    st.raw("s_mov_b32 %s, %%[wave]" % s(S_TMP))
    st.raw("s_mov_b32 %s, %%[smem0]" % s(S_TMP2))
    if drop:
        st.raw("s_mov_b64 %s, %%[mkb]" % s(S_MKRS, 2))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(S_MKRS + 1), s(S_MKRS + 1)))
        st.raw("s_mov_b32 %s, %%[msz]" % s(S_MKRS + 2))
        st.raw("s_mov_b32 %s, 0x00020000" % s(S_MKRS + 3))
        st.raw("s_mov_b32 %s, %%[lqp4]" % s(S_LQP4))
    # buffer sizes
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_KRS + 2), s(S_KL)))                        # K, V: k_len rows of 128 bytes (keys past it read as zeros)
    st.raw("s_mov_b32 %s, %s" % (s(S_VRS + 2), s(S_KRS + 2)))
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_QRS + 2), s(S_LQ)))
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_DORS + 2), s(S_LQ)))
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_DORS + 2), s(S_DORS + 2), s(S_H128)))
    st.raw("s_add_u32 %s, %s, 128" % (s(S_DORS + 2), s(S_DORS + 2)))                # dO, O: the head's 128 bytes of Lq token rows
    st.raw("s_mov_b32 %s, %s" % (s(S_ORS + 2), s(S_DORS + 2)))
    st.raw("s_lshl_b32 %s, %s, 2" % (s(S_LRS + 2), s(S_LQ)))
    st.raw("s_mov_b32 %s, -1" % s(S_WRS + 2))                                        # the workspace: offsets are in range by construction
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_DQRS + 2), s(S_LQ)))
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_DQRS + 2), s(S_DQRS + 2), s(S_LDQ)))
    st.raw("s_add_u32 %s, %s, 128" % (s(S_DQRS + 2), s(S_DQRS + 2)))                # dQ: Lq rows of ldq * 2 bytes, this head's 128 bytes
    st.raw("s_add_u32 %s, %s, 63" % (s(S_NKT), s(S_KL)))
    st.raw("s_lshr_b32 %s, %s, 6" % (s(S_NKT), s(S_NKT)))
    st.raw("s_lshl_b32 %s, %s, 11" % (s(S_KDST), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_KDST), s(S_KDST), s(S_TMP2)))
    st.raw("s_sub_u32 %s, %s, 0x10000" % (s(S_KDST), s(S_KDST)))
    st.raw("s_mul_i32 %s, %s, 9216" % (s(S_STAGE), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_STAGE), s(S_STAGE), s(S_TMP2)))
    st.raw("s_and_b32 %s, %s, 63" % (s(S_TMP), s(S_KL)))                             # the tile that is ragged (none: -1)
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_MASKT), s(S_NKT)))
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_TMP))
    st.raw("s_cselect_b32 %s, -1, %s" % (s(S_MASKT), s(S_MASKT)))
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_MASKT))
    st.raw("s_cselect_b32 %s, 1, 0" % s(S_SPECIAL))
    st.raw("s_mov_b32 %s, %s" % (s(S_REM), s(S_KL)))
    st.raw("s_mov_b32 %s, 0" % s(S_SOFF))
    st.raw("s_mov_b32 %s, 0" % s(S_T))
    st.raw("s_getpc_b64 %s" % s(S_TMP, 2))
    st.label(".Lhere_" + U)
    for reg, lab in ((S_SPEC_A, ".Lspecial_A_"), (S_SPEC_B, ".Lspecial_B_")):
        st.raw("s_add_u32 %s, %s, %s%s-.Lhere_%s" % (s(reg), s(S_TMP), lab, U, U))
        st.raw("s_addc_u32 %s, %s, 0" % (s(reg + 1), s(S_TMP2)))
    # lane-derived values
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(TC))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(TC), v(TC)))                       # lane
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(SH4), v(TC)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(SH4), v(SH4)))                        # 4 hh
    st.raw("v_and_b32_e32 %s, 31, %s" % (v(E1), v(TC)))                             # r
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(E1), s(S_ROW0), v(E1)))                  # this lane's query row
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(MOFF), v(E1)))                        # 4 * row: lse, the workspace and the Mk image
    st.comment("---- tiles 0 1 2 by LDS-DMA; Q, dO, O fragments and lse of this lane's query")
    st.raw("s_nop 4")
    for tile in range(3):
        for p in range(4):
            emit_dma_piece(st, p, tile * SLOT)
            st.raw("s_nop 0")
            emit_dma_load(st, p)
        st.raw("s_add_u32 %s, %s, 0x2000" % (s(S_SOFF), s(S_SOFF)))
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SA + 4 * ks, 4), v(QOFF), s(S_QRS, 4), 32 * ks))
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SB + 4 * ks, 4), v(E0), s(S_DORS, 4), 32 * ks))
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(DPA + 4 * ks, 4), v(E0), s(S_ORS, 4), 32 * ks))
    st.raw("buffer_load_dword %s, %s, %s, 0 offen" % (v(DPB), v(MOFF), s(S_LRS, 4)))
    st.comment("---- state: dQ = 0; dS of half B = 0; the K part of ring slot 3 (the tile 'before' tile 0) = 0")
    for i in range(32):
        st.raw("v_accvgpr_write_b32 %s, 0" % a(DQ + i))
    for r in rng(T, 4):
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    st.raw("v_lshlrev_b32_e32 %s, 4, %s" % (v(TC), v(TC)))                          # 16 * lane
    st.raw("s_add_u32 %s, %s, 0x10000" % (s(S_TMP), s(S_KDST)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(TC), s(S_TMP), v(TC)))
    for p in range(2):
        st.raw("ds_write_b128 %s, %s offset:%d" % (v(TC), v(T, 4), 3 * SLOT + 1024 * p))
    st.raw("s_waitcnt vmcnt(0)")
    for i in range(16):
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(QFR + i), v(SA + i)))
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(DOFR + i), v(SB + i)))
    st.comment("---- delta = rowsum(dO o O) over this lane's 32 d (the other 32 sit on lane ^ 32)")
    st.raw("v_mov_b32_e32 %s, 0" % v(E2))
    for i in range(16):
        st.raw("v_lshlrev_b32_e32 %s, 16, %s" % (v(T), v(SB + i)))
        st.raw("v_lshlrev_b32_e32 %s, 16, %s" % (v(T + 1), v(DPA + i)))
        st.raw("v_and_b32_e32 %s, 0xffff0000, %s" % (v(T + 2), v(SB + i)))
        st.raw("v_and_b32_e32 %s, 0xffff0000, %s" % (v(T + 3), v(DPA + i)))
        st.raw("v_fmac_f32_e32 %s, %s, %s" % (v(E2), v(T), v(T + 1)))
        st.raw("v_fmac_f32_e32 %s, %s, %s" % (v(E2), v(T + 2), v(T + 3)))
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(T))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(T), v(T)))
    st.raw("v_xor_b32_e32 %s, 32, %s" % (v(T), v(T)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(T), v(T)))
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("ds_bpermute_b32 %s, %s, %s" % (v(T + 1), v(T), v(E2)))
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("v_add_f32_e32 %s, %s, %s" % (v(E2), v(E2), v(T + 1)))                   # delta
    # -delta (0 for a padded row), -lse (-inf for a padded row) -> the workspace (lanes 0..31: one row each), and the C operand vectors
    st.raw("v_cmp_gt_i32_e32 vcc, %s, %s" % (s(S_LQ), v(E1)))                       # row < Lq
    st.raw("v_sub_f32_e32 %s, 0, %s" % (v(E2), v(E2)))
    st.raw("v_sub_f32_e32 %s, 0, %s" % (v(E3), v(DPB)))
    st.raw("v_mov_b32_e32 %s, 0xff800000" % v(T + 2))
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(E2), v(E2)))
    st.raw("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(T + 3), v(T + 2), v(E3)))       # what the dK / dV kernel starts from
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(E3), v(E3)))                     # here a padded row just stays finite (it is never stored)
    st.raw("s_add_u32 %s, %s, 63" % (s(S_TMP), s(S_LQ)))
    st.raw("s_and_b32 %s, %s, 0xffffffc0" % (s(S_TMP), s(S_TMP)))                    # Lq padded to whole 64-query tiles: the rows the workspace has
    st.raw("v_cmp_gt_i32_e32 vcc, %s, %s" % (s(S_TMP), v(E1)))
    st.raw("s_and_b32 exec_lo, exec_lo, vcc_lo")                                     # lanes 0..31 hold one row each
    st.raw("s_mov_b32 exec_hi, 0")
    st.raw("buffer_store_dword %s, %s, %s, 0 offen" % (v(E2), v(MOFF), s(S_WRS, 4)))
    st.raw("buffer_store_dword %s, %s, %s, %s offen" % (v(T + 3), v(MOFF), s(S_WRS, 4), s(S_NLOFF)))
    st.raw("s_mov_b32 exec_lo, -1")
    st.raw("s_mov_b32 exec_hi, -1")
    for i in range(16):
        st.raw("v_mov_b32_e32 %s, %s" % (v(NL + i), v(E3)))
        st.raw("v_mov_b32_e32 %s, %s" % (v(ND + i), v(E2)))
    for r in rng(DPB, 16) + rng(T, 4):
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    if drop:
        st.raw("s_mov_b32 %s, 0" % s(S_M0))
        st.raw("s_mov_b32 %s, %s" % (s(S_M1), s(S_LQP4)))
        st.raw("s_nop 2")
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(WKA), v(MOFF), s(S_MKRS, 4), s(S_M0)))
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(WKB), v(MOFF), s(S_MKRS, 4), s(S_M1)))
        st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M0), s(S_LQP4), s(S_M0)))
        st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M1), s(S_LQP4), s(S_M1)))
        st.raw("s_waitcnt vmcnt(0)")
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("s_barrier")
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_NKT))
    st.raw("s_cbranch_scc1 .Lfinal_" + U)
    st.comment("---- (S^T, dP^T) of half A of tile 0; transposed fragments of the (zero) tile -1")
    st.raw("s_nop 7", states=8)
    for j in range(4):
        emit_row_read(st, j, "A", 0)
    for j in range(4):
        emit_sdp(st, "A", j)
    for j in range(4, 8):
        emit_row_read(st, j, "A", 0)
    for j in range(4, 8):
        emit_sdp(st, "A", j)
    for j in range(4):
        emit_tr_reads(st, j, "B", 3 * SLOT)
    # ---- the loop: four tiles per trip ----------------------------------------------------------------------------------------------------
    entry_lds = [set(x) for x in st.lds]
    body_start = len(st.out)
    n_start, mf_start = st.n, dict(st.mfma_w)
    for final_pass in (False, True):
        if final_pass:
            shift = st.n - n_start
            seeded = {r: w - shift for r, w in st.mfma_w.items()}
            if not ABL:
                assert [sorted(x) for x in st.lds] == [sorted(x) for x in entry_lds], "LDS queue at the back edge differs from the entry's"
            del st.out[body_start:]
            st.n = n_start
            st.mfma_w = dict(mf_start)
            for r, w in seeded.items():
                st.mfma_w[r] = max(w, st.mfma_w.get(r, -10 ** 9))
            st.lds = [set(x) for x in entry_lds]
            st.nops = 0
            st.counts = {}
        st.label(".Ltrip_" + U)
        for u in range(4):
            st.comment("==== tile t, t & 3 == %d" % u)
            emit_phase(st, "A", "B", u, "a%d" % u)
            if not (ABL & 16):
                st.raw("s_waitcnt vmcnt(%d)" % (4 + (2 if drop else 0)), kind="wait")
                st.raw("s_barrier", kind="salu")
            emit_phase(st, "B", "A", u, "b%d" % u)
            st.raw("s_add_u32 %s, %s, 1" % (s(S_T), s(S_T)), kind="salu")
            st.raw("s_sub_u32 %s, %s, 64" % (s(S_REM), s(S_REM)), kind="salu")
            if drop:
                st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M0), s(S_LQP4), s(S_M0)), kind="salu")
                st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M1), s(S_LQP4), s(S_M1)), kind="salu")
            st.raw("s_cmp_eq_u32 %s, %s" % (s(S_T), s(S_MASKT)), kind="salu")
            st.raw("s_cselect_b32 %s, 1, 0" % s(S_SPECIAL), kind="salu")
            st.raw("s_cmp_lt_u32 %s, %s" % (s(S_T), s(S_NKT)), kind="salu")
            if u < 3:
                st.raw("s_cbranch_scc0 .Ldone_" + U, kind="salu")
            else:
                st.raw("s_cbranch_scc1 .Ltrip_" + U, kind="salu")
    body_counts = dict(st.counts)
    body_nops = st.nops
    st.label(".Ldone_" + U)
    for j in range(4):
        emit_dq_mfma(st, "B", j)
    st.label(".Lfinal_" + U)
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    st.raw("s_nop 7", states=8)
    st.raw("s_nop 7", states=8)
    st.comment("---- epilogue: dQ^T * scale -> bf16, through LDS (rows of 144 bytes) to whole-row stores")
    st.raw("s_barrier")
    LN, RR, WA, RA, SO, XA = T, T + 1, T + 2, T + 3, T + 4, T + 5
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(LN))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(LN), v(LN)))
    st.raw("v_and_b32_e32 %s, 31, %s" % (v(RR), v(LN)))
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(WA), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 3, %s" % (v(WA), v(WA)))
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(WA), v(RR), v(WA)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(WA), v(RR), v(WA)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(WA), s(S_STAGE), v(WA)))
    st.raw("v_lshrrev_b32_e32 %s, 3, %s" % (v(RA), v(LN)))
    st.raw("v_and_b32_e32 %s, 7, %s" % (v(SO), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 4, %s" % (v(SO), v(SO)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(XA), s(S_ROW0), v(RA)))
    st.raw("v_mul_lo_u32 %s, %s, %s" % (v(XA), v(XA), s(S_LDQ)))
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(T + 6), v(RA), v(SO)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(RA), v(RA), v(T + 6)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(RA), s(S_STAGE), v(RA)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(SO), v(XA), v(SO)))
    R = SA
    for cb in range(2):
        for g in range(4):
            for j in range(4):
                st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + j), a(DQ + 16 * cb + 4 * g + j)))
            st.raw("s_nop 0")
            for j in range(4):
                st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + j), s(S_SCALE), v(R + j)))
            st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 4), v(R), v(R + 1)))
            st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 5), v(R + 2), v(R + 3)))
            st.raw("ds_write_b64 %s, %s offset:%d" % (v(WA), v(R + 4, 2), 64 * cb + 16 * g))
    st.raw("s_waitcnt lgkmcnt(0)")
    for i in range(4):
        st.raw("ds_read_b128 %s, %s offset:%d" % (v(SA + 4 * i, 4), v(RA), i * 8 * 144))
    st.raw("s_lshl_b32 %s, %s, 3" % (s(S_TMP2), s(S_LDQ)))
    st.raw("s_mov_b32 %s, 0" % s(S_TMP))
    for i in range(4):
        st.raw("s_waitcnt lgkmcnt(%d)" % (3 - i))
        st.raw("buffer_store_dwordx4 %s, %s, %s, %s offen" % (v(SA + 4 * i, 4), v(SO), s(S_DQRS, 4), s(S_TMP)))
        st.raw("s_add_u32 %s, %s, %s" % (s(S_TMP), s(S_TMP), s(S_TMP2)))
    st.raw("s_endpgm")
    emit_special(st, "A")
    emit_special(st, "B")
    return st, body_counts, body_nops


def main():
    global ABL
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "end-to-end_asr_pytorch_amd", "csrc", "attention_bwd_dq_asm.inc")
    args = sys.argv[1:]
    while args and args[0].startswith("--"):
        if args[0] == "--abl":
            ABL = int(args[1])
        elif args[0] == "--out":
            out = args[1]
        args = args[2:]
    with open(out, "w") as f:
        f.write("// generated by tools/gen_attn_bwd_dq.py - do not edit (edit the generator and run it again)\n")
        for drop in (False, True):
            st, counts, nops = build(drop)
            f.write("#define ATTN_BWD_DQ_ASM_%s \\\n" % ("TRAIN" if drop else "EVAL"))
            for line in st.out:
                f.write('    "%s\\n" \\\n' % line.replace("\\", "\\\\").replace('"', '\\"'))
            f.write('    ""\n')
            sys.stderr.write("dq %s: %d lines; per trip of 4 tiles: %s; s_nop states padded in the loop: %d\n" %
                             ("train" if drop else "eval", len(st.out), counts, nops))
        regs = ["v%d" % i for i in list(range(64)) + list(range(96, 128))] + ["a%d" % i for i in range(96)] + ["s%d" % i for i in range(34, 100)] + ["vcc", "memory"]
        f.write("#define ATTN_BWD_DQ_ASM_CLOBBERS %s\n" % ", ".join('"%s"' % r for r in regs))


if __name__ == "__main__":
    main()
