#!/usr/bin/env python3
"""Generator of the attention backward's dK / dV instruction stream (csrc/attention_bwd_asm.inc, included by attention_bwd4.hip).

Same method as tools/gen_attn_fwd4.py (every instruction of a phase assigned to an MFMA gap, LDS queue counted, MFMA -> VALU distances
padded), for the key-stationary half of the flash backward (attention.py:76-84 differentiated):

  a wave owns 32 keys (key on the lane); K and V fragments stay in AGPRs; dK^T and dV^T [64 d x 32 keys] accumulate in AGPRs;
  the workgroup (4 waves = 128 keys) walks the queries in tiles of 64 = two halves A, B of 32 queries:
      S  = Q.K^T - lse      (A = Q rows from LDS, B = K fragments; the accumulator STARTS at -lse read from LDS: no VALU subtract)
      dP = dO.V^T - delta   (same with dO, V, -delta; with dropout the accumulator starts at 0 and the mask is applied first)
      P  = exp2(S), dS = P o dP;   dV^T += dO^T.P,  dK^T += Q^T.dS   (A = transposed reads of the same dO / Q tiles, B = P / dS packed
      to bf16 IN PLACE over the dead halves of the S / dP accumulators)
  phase X: the vector port turns half X's (S, dP) into (P, dS) while the matrix pipe runs the OTHER half's two outer products (8 MFMAs)
  and its next tile's S and dP (8 MFMAs).  One barrier and one counted vmcnt per tile; Q, dO, -lse, -delta tiles arrive by LDS-DMA three
  tiles ahead into a ring of four slots.  Two waves per SIMD: 128 VGPRs + 128 AGPRs (fragment buffers live in AGPRs, LDS reads write them
  directly).
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import gen_attn_fwd4 as F
from gen_attn_fwd4 import Stream, v, a, s, rng

DROP = False
ABL = 0       # timing-only builds: 1 no MFMA, 2 no VALU, 4 no fragment reads, 8 no LDS-DMA, 16 no barrier

# ---- registers ---------------------------------------------------------------------------------------------------------------------
SA, SB, DPA, DPB = 0, 16, 32, 48      # S and dP accumulators of the two halves (16 registers each); their low 8 become packed P / dS
DB = 64                               # -delta of the half the vector port works on (dropout only)
T = 80                                # 8 rotating temporaries
RB, TB = 96, 100                      # LDS read addresses: row fragments per k-step; transposed fragments [d half][lo / hi]
LSB = 104                             # -lse / -delta read address
VOFFQ, VOFFD = 105, 107               # LDS-DMA source offsets of this lane's two pieces (Q rows of 128 bytes; dO rows of h * 128)
WQA, WQB, SH4, TC, TD, MOFF, SMOFF, KFOFF = 109, 110, 111, 112, 113, 114, 115, 116     # (WQA + 8, WQB + 8: the words of the next tile)
E0 = 119                              # epilogue / scratch 119..127
DK, DV, KFR, VFR, RF, TF = 0, 32, 64, 80, 96, 112      # AGPRs
S_QRS, S_DORS, S_KRS, S_VRS, S_SMRS, S_DKRS, S_DVRS, S_MQRS = 36, 40, 44, 48, 52, 56, 60, 64
S_T, S_NQT, S_KL, S_LQ, S_H128, S_LDKV, S_KDST, S_SDST = 68, 69, 70, 71, 72, 73, 74, 75
S_QSOFF, S_DSOFF, S_SSOFF, S_TMP, S_TMP2, S_DSC, S_KEY0, S_STAGE, S_MQOFF, S_LKP4, S_LK = 76, 77, 78, 79, 80, 81, 82, 83, 84, 85, 86
SLOT = 16384                          # LDS: [4 x 512 bytes: -lse 256 | -delta 256 of a tile][4 x 16384: Q tile 8192 | dO tile 8192]
SSLOT = 512                           # (the tile ring starts at byte 2048: the read bases carry that, every offset stays below 65536)


def half(X):
    return dict(S=SA, DP=DPA, WQ=WQA, H=0) if X == "A" else dict(S=SB, DP=DPB, WQ=WQB, H=1)


def lds_read(st, txt, addr, awrites=(), vwrites=()):
    """an LDS read whose destination is AGPRs (fragment buffers; tracked as 1000 + n) or VGPRs."""
    if ABL & 4:
        return
    st._need(set(addr))
    st._mfma_pad(set(addr) | set(vwrites))
    st.raw(txt, kind="lds")
    st.lds.append(set(1000 + r for r in awrites) | set(vwrites))


def mfma(st, txt, areads=(), vreads=(), vwrites=()):
    st._need(set(1000 + r for r in areads) | set(vreads) | set(vwrites))
    if ABL & 1:
        return
    st._mfma_pad(set(vreads))
    st.raw(txt, kind="mfma")
    for r in vwrites:
        st.mfma_w[r] = st.n - 1


# ---- MFMAs of a phase (matrix side works on half Y) ---------------------------------------------------------------------------------
def emit_outprod(st, Y, j):
    """j = 0..3: dV, 4..7: dK; query-step g' = (j >> 1) & 1, d half cb = j & 1.  A = transposed fragment in TF slot j & 3."""
    y = half(Y)
    which, g, cb = j >> 2, (j >> 1) & 1, j & 1
    acc = (DV if which == 0 else DK) + 16 * cb
    b = (y["S"] if which == 0 else y["DP"]) + 4 * g
    tf = TF + 4 * (j & 3)
    mfma(st, "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (a(acc, 16), a(tf, 4), v(b, 4), a(acc, 16)), rng(tf, 4), rng(b, 4))


def emit_sdp(st, Y, j):
    """j = 0..7: S k-steps 0..3, then dP k-steps 0..3.  A = row fragment in RF slot j & 3."""
    y = half(Y)
    ks, which = j & 3, j >> 2
    acc = y["S"] if which == 0 else y["DP"]
    bfr = (KFR if which == 0 else VFR) + 4 * ks
    rf = RF + 4 * (j & 3)
    c = v(acc, 16)
    if which == 1 and ks == 0 and DROP:
        c = "0"
    mfma(st, "v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(acc, 16), a(rf, 4), a(bfr, 4), c), rng(rf, 4), [], rng(acc, 16))


def emit_tr_reads(st, j, Y, slot_off, base=TB):
    """transposed fragment of out-product j of half Y: dO^T (dV, j < 4) or Q^T (dK), query-step g', d half cb."""
    which, g, cb = j >> 2, (j >> 1) & 1, j & 1
    tf = TF + 4 * (j & 3)
    off = slot_off + (8192 if which == 0 else 0) + (2 * half(Y)["H"] + g) * 2048
    lds_read(st, "ds_read_b64_tr_b16 %s, %s offset:%d" % (a(tf, 2), v(base + 2 * cb), off), [base + 2 * cb], rng(tf, 2))
    lds_read(st, "ds_read_b64_tr_b16 %s, %s offset:%d" % (a(tf + 2, 2), v(base + 2 * cb + 1), off), [base + 2 * cb + 1], rng(tf + 2, 2))


def emit_row_read(st, j, Y, slot_off):
    ks, which = j & 3, j >> 2
    rf = RF + 4 * (j & 3)
    off = slot_off + (8192 if which == 1 else 0) + 4096 * half(Y)["H"]
    lds_read(st, "ds_read_b128 %s, %s offset:%d" % (a(rf, 4), v(RB + ks), off), [RB + ks], rng(rf, 4))


def emit_cinit_read(st, Y, which, g, slot_off):
    """accumulator start values of half Y: -lse (which = 0) or -delta (1) of queries 8g + 4hh .. + 3 (registers 4g .. 4g + 3)."""
    y = half(Y)
    dst = (y["S"] if which == 0 else y["DP"]) + 4 * g
    off = (slot_off // SLOT) * SSLOT + 256 * which + 128 * y["H"] + 32 * g
    lds_read(st, "ds_read_b128 %s, %s offset:%d" % (v(dst, 4), v(LSB), off), [LSB], [], rng(dst, 4))


def emit_db_read(st, X, g, slot_off):
    x = half(X)
    off = (slot_off // SLOT) * SSLOT + 256 + 128 * x["H"] + 32 * g
    lds_read(st, "ds_read_b128 %s, %s offset:%d" % (v(DB + 4 * g, 4), v(LSB), off), [LSB], [], rng(DB + 4 * g, 4))


# ---- vector side: element i of half X ----------------------------------------------------------------------------------------------------
def emit_exp(st, X, i):
    if ABL & 2:
        return
    x = half(X)
    ta = T + (2 * i) % 8
    st.valu("v_exp_f32_e32 %s, %s" % (v(ta), v(x["S"] + i)), [x["S"] + i], [ta], kind="exp")


def emit_elem(st, X, i, wq=None):
    """P and dS of element i (its exp was issued a gap earlier); on odd i the pair is packed in place."""
    if ABL & 2:
        return
    x = dict(half(X))
    if wq is not None:
        x["WQ"] = wq
    ta, tb = T + (2 * i) % 8, T + (2 * i + 1) % 8
    dp = x["DP"] + i
    if DROP:
        st.valu("v_bfe_i32 %s, %s, %d, 1" % (v(TC), v(x["WQ"]), 8 * (i >> 2) + (i & 3)), [x["WQ"]], [TC], kind="drop")
        st.valu("v_and_b32_e32 %s, %s, %s" % (v(tb), v(ta), v(TC)), [ta, TC], [tb], kind="drop")            # the kept probability (dV)
        st.valu("v_and_b32_e32 %s, %s, %s" % (v(dp), v(dp), v(TC)), [dp, TC], [dp], kind="drop")            # dropout's mask on dP ...
        st.valu("v_fma_f32 %s, %s, %s, %s" % (v(dp), v(dp), s(S_DSC), v(DB + i)), [dp, DB + i], [dp], kind="drop")     # ... / keep - delta
    st.valu("v_mul_f32_e32 %s, %s, %s" % (v(dp), v(ta), v(dp)), [ta, dp], [dp])                              # dS = P (dP - delta)
    if i & 1:
        k = i >> 1
        p0 = T + ((2 * (i - 1) + 1) % 8 if DROP else (2 * (i - 1)) % 8)
        p1 = tb if DROP else ta
        st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["S"] + k), v(p0), v(p1)), [p0, p1], [x["S"] + k], kind="cvt")
        st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["DP"] + k), v(dp - 1), v(dp)), [dp - 1, dp], [x["DP"] + k], kind="cvt")


def emit_phase(st, X, Y, u, uid):
    """vector port: half X of tile t -> (P, dS).  matrix pipe: out-products of half Y (tile t-1 for Y = B, t for Y = A), then S and dP of
    half Y's next tile (t for Y = B, t+1 for Y = A).  u = t & 3.  Phase B also carries this tile's five LDS-DMA requests, one every third gap
    (issued in one burst behind the barrier they held the wave for ~100 cycles each)."""
    x, y = half(X), half(Y)
    st.comment("---- phase %s: (P, dS) of half %s | out-products and next (S, dP) of half %s" % (uid, X, Y))
    if X == "A":
        s_out, s_nxt, s_pre, s_db = (u + 3) & 3, u, u, u            # out-products of B(t-1); S/dP of B(t); next phase's fragments: A(t); delta of B(t)
    else:
        s_out, s_nxt, s_pre, s_db = u, (u + 1) & 3, u, (u + 1) & 3   # out-products of A(t); S/dP of A(t+1); next: B(t); delta of A(t+1)
    wq = x["WQ"] + 8 * (u & 1)            # the keep bits of this tile's half (two registers per half, alternating by tile)
    if DROP:         # requested a step ahead; this lane's queries are bits 8 g + 4 hh + x; then the request of the next tile's word
        st.raw("s_waitcnt vmcnt(6)", kind="wait")
        st.raw("v_lshrrev_b32_e32 %s, %s, %s" % (v(wq), v(SH4), v(wq)), kind="drop")
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(x["WQ"] + 8 * ((u + 1) & 1)), v(MOFF), s(S_MQRS, 4), s(S_MQOFF)), kind="mload")
        st.raw("s_add_u32 %s, %s, %s" % (s(S_MQOFF), s(S_MQOFF), s(S_LKP4)), kind="salu")
    dma_at = {2: 0, 5: 1, 8: 2, 11: 3, 14: 4} if (X == "B" and not (ABL & 8)) else {}
    for gap in range(16):
        if gap < 8:
            emit_outprod(st, Y, gap)
        else:
            emit_sdp(st, Y, gap - 8)
        # LDS reads of this gap
        if gap < 4:
            emit_tr_reads(st, gap + 4, Y, s_out * SLOT)              # transposed fragments 4..7 of this phase's out-products
        elif gap < 12:
            emit_row_read(st, gap - 4, Y, s_nxt * SLOT)              # row fragments of the S / dP MFMAs, four gaps ahead
        else:
            emit_tr_reads(st, gap - 12, X, s_pre * SLOT)             # transposed fragments 0..3 of the NEXT phase's out-products (half X)
        # accumulator start values of half Y's next (S, dP): a register quad only after the out-product that reads it as packed P / dS
        for (g0, wh, g) in ((0, 0, 2), (0, 0, 3), (1, 1, 2), (1, 1, 3), (2, 0, 0), (4, 0, 1), (6, 1, 0), (8, 1, 1)):
            if gap == g0 and not (wh == 1 and DROP):
                emit_cinit_read(st, Y, wh, g, s_nxt * SLOT)
        if DROP and gap in (5, 9, 13):                                # -delta of the half the vector port takes NEXT (its registers free up in groups)
            emit_db_read(st, Y, (gap - 5) // 4, s_db * SLOT)
        if gap in dma_at:
            emit_dma_m0(st, dma_at[gap], ((u + 3) & 3) * SLOT)
        emit_exp(st, X, gap)
        if gap in dma_at:
            if ABL & 2:
                st.raw("s_nop 0", kind="nop")
            emit_dma_load(st, dma_at[gap])
        if gap >= 1:
            emit_elem(st, X, gap - 1, wq)
    emit_elem(st, X, 15, wq)
    if DROP:
        emit_db_read(st, Y, 3, s_db * SLOT)
    if dma_at:
        emit_dma_advance(st)


def emit_dma_m0(st, p, slot_imm):
    """request p of a tile's five: Q pieces 0, 1; dO pieces 2, 3; 4 = the 256-byte run of -lse or -delta."""
    if p < 4:
        st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_KDST), slot_imm + 8192 * (p >> 1) + (p & 1) * 1024 + 0x10000), kind="salu")
    else:
        st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_SDST), (slot_imm // SLOT) * SSLOT + 0x10000), kind="salu")


def emit_dma_load(st, p):
    if p < 4:
        st.raw("buffer_load_dwordx4 %s, %s, %s offen lds" % (v((VOFFQ if p < 2 else VOFFD) + (p & 1)), s(S_QRS if p < 2 else S_DORS, 4),
                                                              s(S_QSOFF if p < 2 else S_DSOFF)), kind="dma")
    else:
        st.raw("buffer_load_dword %s, %s, %s offen lds" % (v(SMOFF), s(S_SMRS, 4), s(S_SSOFF)), kind="dma")


def emit_dma_advance(st):
    st.raw("s_add_u32 %s, %s, 0x2000" % (s(S_QSOFF), s(S_QSOFF)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 6" % (s(S_TMP), s(S_H128)), kind="salu")
    st.raw("s_add_u32 %s, %s, %s" % (s(S_DSOFF), s(S_DSOFF), s(S_TMP)), kind="salu")
    st.raw("s_add_u32 %s, %s, 0x100" % (s(S_SSOFF), s(S_SSOFF)), kind="salu")


def emit_dma_group(st, u_slot_imm=None, dyn=None):
    """one tile's requests of this wave in one go (prologue): 2 Q pieces, 2 dO pieces, one 256-byte run of -lse or -delta."""
    if ABL & 8:
        return
    for p in range(5):
        emit_dma_m0(st, p, u_slot_imm)
        st.raw("s_nop 0", kind="nop")
        emit_dma_load(st, p)
    emit_dma_advance(st)


def build(drop):
    global DROP
    DROP = drop
    st = Stream()
    U = "%="
    st.comment("==== attention backward dK / dV (%s): generated by tools/gen_attn_bwd.py - do not edit" % ("train: dropout" if drop else "eval"))
    # ---- inputs -----------------------------------------------------------------------------------------------------------------------
    st.raw("v_and_b32_e32 %s, 0xffff, %%[voffq]" % v(VOFFQ))
    st.raw("v_lshrrev_b32_e32 %s, 16, %%[voffq]" % v(VOFFQ + 1))
    st.raw("v_mov_b32_e32 %s, %%[voffd0]" % v(VOFFD))
    st.raw("v_mov_b32_e32 %s, %%[voffd1]" % v(VOFFD + 1))
    for i, nm in enumerate(("rb01", "rb23")):
        st.raw("v_and_b32_e32 %s, 0xffff, %%[%s]" % (v(RB + 2 * i), nm))
        st.raw("v_lshrrev_b32_e32 %s, 16, %%[%s]" % (v(RB + 2 * i + 1), nm))
    for i, nm in enumerate(("tb01", "tb23")):
        st.raw("v_and_b32_e32 %s, 0xffff, %%[%s]" % (v(TB + 2 * i), nm))
        st.raw("v_lshrrev_b32_e32 %s, 16, %%[%s]" % (v(TB + 2 * i + 1), nm))
    st.raw("v_mov_b32_e32 %s, %%[kfoff]" % v(KFOFF))
    for name, reg in (("qb", S_QRS), ("dob", S_DORS), ("kb", S_KRS), ("vb", S_VRS), ("smb", S_SMRS), ("dkb", S_DKRS), ("dvb", S_DVRS)):
        st.raw("s_mov_b64 %s, %%[%s]" % (s(reg, 2), name))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(reg + 1), s(reg + 1)))
        st.raw("s_mov_b32 %s, 0x00020000" % s(reg + 3))
    for name, reg in (("kl", S_KL), ("lq", S_LQ), ("lk", S_LK), ("h128", S_H128), ("ldkv2", S_LDKV), ("dsc", S_DSC), ("key0", S_KEY0)):
        st.raw("s_mov_b32 %s, %%[%s]" % (s(reg), name))
    st.raw("s_mov_b32 %s, %%[wave]" % s(S_TMP))
    st.raw("s_mov_b32 %s, %%[smem0]" % s(S_TMP2))
    if drop:
        st.raw("s_mov_b64 %s, %%[mqb]" % s(S_MQRS, 2))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(S_MQRS + 1), s(S_MQRS + 1)))
        st.raw("s_mov_b32 %s, %%[msz]" % s(S_MQRS + 2))
        st.raw("s_mov_b32 %s, 0x00020000" % s(S_MQRS + 3))
        st.raw("s_mov_b32 %s, %%[lkp4]" % s(S_LKP4))
    # lane-derived values
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(TC))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(TC), v(TC)))                       # lane
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(SMOFF), v(TC)))                       # 4 * lane: the -lse / -delta run
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(SH4), v(TC)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(SH4), v(SH4)))                        # 4 hh
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(LSB), v(SH4)))                        # 16 hh
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(LSB), s(S_TMP2), v(LSB)))               # + smem0 (slot, half and quad go into the offsets)
    # buffer sizes
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_QRS + 2), s(S_LQ)))                        # Q: Lq rows of 128 bytes
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_DORS + 2), s(S_LQ)))
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_DORS + 2), s(S_DORS + 2), s(S_H128)))
    st.raw("s_add_u32 %s, %s, 128" % (s(S_DORS + 2), s(S_DORS + 2)))                # dO: the head's 128 bytes of Lq token rows
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_KRS + 2), s(S_KL)))                        # K, V: k_len rows (keys past it read as zeros and are stored as zeros)
    st.raw("s_mov_b32 %s, %s" % (s(S_VRS + 2), s(S_KRS + 2)))
    st.raw("s_add_u32 %s, %s, 63" % (s(S_NQT), s(S_LQ)))
    st.raw("s_lshr_b32 %s, %s, 6" % (s(S_NQT), s(S_NQT)))
    st.raw("s_lshl_b32 %s, %s, 8" % (s(S_SMRS + 2), s(S_NQT)))                      # -lse / -delta: whole tiles of 64 floats
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_DKRS + 2), s(S_LK)))
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_DKRS + 2), s(S_DKRS + 2), s(S_LDKV)))
    st.raw("s_add_u32 %s, %s, 128" % (s(S_DKRS + 2), s(S_DKRS + 2)))                # dK / dV: Lk rows of ldkv * 2 bytes, this head's 128 bytes
    st.raw("s_mov_b32 %s, %s" % (s(S_DVRS + 2), s(S_DKRS + 2)))
    # LDS destinations of this wave's requests (biased by -0x10000: keeps every immediate of the M0 sums positive)
    st.raw("s_lshl_b32 %s, %s, 11" % (s(S_KDST), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_KDST), s(S_KDST), s(S_TMP2)))
    st.raw("s_sub_u32 %s, %s, 0x%x" % (s(S_KDST), s(S_KDST), 0x10000 - 2048))
    st.raw("s_and_b32 %s, %s, 1" % (s(S_SDST), s(S_TMP)))
    st.raw("s_lshl_b32 %s, %s, 8" % (s(S_SDST), s(S_SDST)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_SDST), s(S_SDST), s(S_TMP2)))
    st.raw("s_sub_u32 %s, %s, 0x10000" % (s(S_SDST), s(S_SDST)))
    st.raw("s_mul_i32 %s, %s, 9216" % (s(S_STAGE), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_STAGE), s(S_STAGE), s(S_TMP2)))            # the epilogue's 2 x 32 rows of 144 bytes
    st.raw("s_mov_b32 %s, 0" % s(S_QSOFF))
    st.raw("s_mov_b32 %s, 0" % s(S_DSOFF))
    st.raw("s_mov_b32 %s, 0" % s(S_SSOFF))
    st.raw("s_mov_b32 %s, 0" % s(S_T))
    st.comment("---- K and V fragments of this lane's key (B operands of every S / dP), tiles 0 1 2 by LDS-DMA")
    st.raw("s_nop 4")
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SA + 4 * ks, 4), v(KFOFF), s(S_KRS, 4), 32 * ks))
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SB + 4 * ks, 4), v(KFOFF), s(S_VRS, 4), 32 * ks))
    if drop:
        st.raw("v_lshrrev_b32_e32 %s, 7, %s" % (v(MOFF), v(KFOFF)))                 # key row * 128 + 16 hh  ->  4 * key
        st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(MOFF), v(MOFF)))
        st.raw("s_mov_b32 %s, 0" % s(S_MQOFF))
        st.raw("s_nop 2")
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(WQA), v(MOFF), s(S_MQRS, 4), s(S_MQOFF)))
        st.raw("s_add_u32 %s, %s, %s" % (s(S_MQOFF), s(S_MQOFF), s(S_LKP4)))
        st.raw("s_nop 2")
        st.raw("buffer_load_dword %s, %s, %s, %s offen" % (v(WQB), v(MOFF), s(S_MQRS, 4), s(S_MQOFF)))
        st.raw("s_add_u32 %s, %s, %s" % (s(S_MQOFF), s(S_MQOFF), s(S_LKP4)))
    for tile in range(3):
        emit_dma_group(st, u_slot_imm=tile * SLOT)
    st.comment("---- state: dK = dV = 0; P and dS of half B = 0; ring slot 3 (the tile 'before' tile 0) = 0")
    for i in range(64):
        st.raw("v_accvgpr_write_b32 %s, 0" % a(i))
    for r in rng(DPA, 32) + rng(T, 4):
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(TC), v(SMOFF)))                       # 16 * lane
    st.raw("s_add_u32 %s, %s, 0x10000" % (s(S_TMP), s(S_KDST)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(TC), s(S_TMP), v(TC)))
    for which in range(2):
        for p in range(2):
            st.raw("ds_write_b128 %s, %s offset:%d" % (v(TC), v(T, 4), 3 * SLOT + 8192 * which + 1024 * p))
    st.raw("s_waitcnt vmcnt(15)")                                                    # K, V fragments (and the first two mask words)
    for i in range(16):
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(KFR + i), v(SA + i)))
    for i in range(16):
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(VFR + i), v(SB + i)))
    for r in rng(SB, 16):
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    st.raw("s_waitcnt vmcnt(10) lgkmcnt(0)")                                         # tile 0; the zero fill
    st.raw("s_barrier")
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_NQT))
    st.raw("s_cbranch_scc1 .Lfinal_" + U)
    st.comment("---- (S, dP) of half A of tile 0; transposed fragments 0..3 of the (zero) tile -1; -delta of half A")
    st.raw("s_nop 7", states=8)
    for g in range(4):
        emit_cinit_read(st, "A", 0, g, 0)
        if not drop:
            emit_cinit_read(st, "A", 1, g, 0)
    for j in range(4):
        emit_row_read(st, j, "A", 0)
    for j in range(4):
        emit_sdp(st, "A", j)
    for j in range(4, 8):
        emit_row_read(st, j, "A", 0)
    for j in range(4, 8):
        emit_sdp(st, "A", j)
    # the reads in flight at the loop's entry, in the order the end of a phase B leaves them (the counted waits assume it)
    if drop:
        emit_db_read(st, "A", 0, 0)
        emit_db_read(st, "A", 1, 0)
        st.drain_lds()
    emit_tr_reads(st, 0, "B", 3 * SLOT)
    emit_tr_reads(st, 1, "B", 3 * SLOT)
    if drop:
        emit_db_read(st, "A", 2, 0)
    emit_tr_reads(st, 2, "B", 3 * SLOT)
    emit_tr_reads(st, 3, "B", 3 * SLOT)
    if drop:
        emit_db_read(st, "A", 3, 0)
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
                st.raw("s_waitcnt vmcnt(%d)" % (5 + (2 if drop else 0)), kind="wait")     # tile t+1 has landed: everything but the last group (and 2 mask words)
                st.raw("s_barrier", kind="salu")
            emit_phase(st, "B", "A", u, "b%d" % u)
            st.raw("s_add_u32 %s, %s, 1" % (s(S_T), s(S_T)), kind="salu")
            st.raw("s_cmp_lt_u32 %s, %s" % (s(S_T), s(S_NQT)), kind="salu")
            if u < 3:
                st.raw("s_cbranch_scc0 .Ldone_" + U, kind="salu")
            else:
                st.raw("s_cbranch_scc1 .Ltrip_" + U, kind="salu")
    body_counts = dict(st.counts)
    body_nops = st.nops
    # ---- after the last tile: half B's out-products; its transposed fragments 4..7 come from slot (nqt - 1) & 3 -----------------------------
    st.label(".Ldone_" + U)
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_TMP), s(S_NQT)))
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_TMP)))
    st.raw("s_mul_i32 %s, %s, %d" % (s(S_TMP), s(S_TMP), SLOT))
    for i in range(4):
        st.raw("v_add_u32_e32 %s, %s, %s" % (v(T + i), s(S_TMP), v(TB + i)))
    for j in range(4):
        emit_outprod(st, "B", j)
    st.raw("s_nop 3", states=4)
    for j in range(4, 8):
        emit_tr_reads(st, j, "B", 0, base=T)
    for j in range(4, 8):
        emit_outprod(st, "B", j)
    st.label(".Lfinal_" + U)
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    st.raw("s_nop 7", states=8)
    st.raw("s_nop 7", states=8)
    st.comment("---- epilogue: dK^T * ln 2 and dV^T (/ keep) -> bf16, through LDS (rows of 144 bytes) to whole-row stores; keys past k_len get zeros")
    st.raw("s_barrier")
    LN, RR, WA, RA, SO, KV = E0, E0 + 1, E0 + 2, E0 + 3, E0 + 4, E0 + 5
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(LN))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(LN), v(LN)))
    st.raw("v_and_b32_e32 %s, 31, %s" % (v(RR), v(LN)))                                  # the lane's key inside the wave's 32
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(WA), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 3, %s" % (v(WA), v(WA)))                                # 8 hh
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(WA), v(RR), v(WA)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(WA), v(RR), v(WA)))                      # + 144 r
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(WA), s(S_STAGE), v(WA)))
    st.raw("v_lshrrev_b32_e32 %s, 3, %s" % (v(RA), v(LN)))                                # row of the read-back: lane >> 3 (+ 8 per instruction)
    st.raw("v_and_b32_e32 %s, 7, %s" % (v(SO), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 4, %s" % (v(SO), v(SO)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(KV), s(S_KEY0), v(RA)))                      # key of the read-back row
    st.raw("v_mul_lo_u32 %s, %s, %s" % (v(KV), v(KV), s(S_LDKV)))
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(E0 + 6), v(RA), v(SO)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(RA), v(RA), v(E0 + 6)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(RA), s(S_STAGE), v(RA)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(SO), v(KV), v(SO)))                          # output byte offset of (key row, chunk)
    # this lane's key valid?  (both of its accumulator columns belong to key0 + r)
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(KV), s(S_KEY0), v(RR)))
    st.raw("v_cmp_gt_i32_e32 vcc, %s, %s" % (s(S_KL), v(KV)))
    st.raw("v_mov_b32_e32 %s, 0x3f317218" % v(E0 + 6))                                   # ln 2: dS is per natural-log score, q carries log2(e)
    st.raw("v_mov_b32_e32 %s, %s" % (v(E0 + 7), s(S_DSC)) if drop else "v_mov_b32_e32 %s, 1.0" % v(E0 + 7))
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(E0 + 6), v(E0 + 6)))
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(E0 + 7), v(E0 + 7)))
    R = SA
    for oi, (acc, scale) in enumerate(((DK, E0 + 6), (DV, E0 + 7))):
        for cb in range(2):
            for g in range(4):
                for j in range(4):
                    st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + j), a(acc + 16 * cb + 4 * g + j)))
                st.raw("s_nop 0")
                for j in range(4):
                    # (a key past k_len may hold inf / NaN: select, do not multiply by 0)
                    st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + j), v(R + j), v(scale)))
                    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(R + j), v(R + j)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 4), v(R), v(R + 1)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 5), v(R + 2), v(R + 3)))
                st.raw("ds_write_b64 %s, %s offset:%d" % (v(WA), v(R + 4, 2), oi * 32 * 144 + 64 * cb + 16 * g))
    st.raw("s_waitcnt lgkmcnt(0)")
    for oi in range(2):
        for i in range(4):
            st.raw("ds_read_b128 %s, %s offset:%d" % (v(SA + 4 * (4 * oi + i), 4), v(RA), oi * 32 * 144 + i * 8 * 144))
    st.raw("s_lshl_b32 %s, %s, 3" % (s(S_TMP2), s(S_LDKV)))                            # 8 key rows
    for oi, rs in enumerate((S_DKRS, S_DVRS)):
        st.raw("s_mov_b32 %s, 0" % s(S_TMP))
        for i in range(4):
            st.raw("s_waitcnt lgkmcnt(%d)" % (7 - (4 * oi + i)))
            st.raw("buffer_store_dwordx4 %s, %s, %s, %s offen" % (v(SA + 4 * (4 * oi + i), 4), v(SO), s(rs, 4), s(S_TMP)))
            st.raw("s_add_u32 %s, %s, %s" % (s(S_TMP), s(S_TMP), s(S_TMP2)))
    st.raw("s_endpgm")
    return st, body_counts, body_nops


def main():
    global ABL
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "end-to-end_asr_pytorch_amd", "csrc", "attention_bwd_asm.inc")
    args = sys.argv[1:]
    while args and args[0].startswith("--"):
        if args[0] == "--abl":
            ABL = int(args[1])
        elif args[0] == "--out":
            out = args[1]
        args = args[2:]
    F.ABL = 0
    with open(out, "w") as f:
        f.write("// generated by tools/gen_attn_bwd.py - do not edit (edit the generator and run it again)\n")
        for drop in (False, True):
            st, counts, nops = build(drop)
            f.write("#define ATTN_BWD_DKV_ASM_%s \\\n" % ("TRAIN" if drop else "EVAL"))
            for line in st.out:
                f.write('    "%s\\n" \\\n' % line.replace("\\", "\\\\").replace('"', '\\"'))
            f.write('    ""\n')
            sys.stderr.write("dkv %s: %d lines; per trip of 4 tiles: %s; s_nop states padded in the loop: %d\n" %
                             ("train" if drop else "eval", len(st.out), counts, nops))
        regs = ["v%d" % i for i in list(range(64)) + list(range(96, 128))] + ["a%d" % i for i in range(128)] + ["s%d" % i for i in range(34, 100)] + ["vcc", "memory"]
        f.write("#define ATTN_BWD_ASM_CLOBBERS %s\n" % ", ".join('"%s"' % r for r in regs))


if __name__ == "__main__":
    main()
