#!/usr/bin/env python3
"""Generator of the encoder self-attention forward's instruction stream (csrc/attention_fwd4_asm.inc, included by attention_fwd4.hip).

Why generated assembly: at d_k = 64 the loop is bound by vector-instruction ISSUE (32 v_exp_f32 + 32 v_add_f32 + 16 v_cvt_pk per
16 MFMAs on one SIMD), so every MFMA gap has to carry its share of the softmax, its LDS fragment reads and its counted waits - hipcc
bunches the MFMAs and the VALU block instead (80 us against v3's 54), with sched_group_barrier it spills or moves the accumulators to
AGPRs and pays a v_accvgpr_read per score.  This script assigns every instruction of a phase to an MFMA gap, counts the LDS queue for the
s_waitcnt lgkmcnt(N) in front of each MFMA, and pads MFMA -> VALU read distances (12 wait states, 8-pass XDL).

Measured on the way (MI355X, [32,4,1000,1000]): the same stream at ONE wave per SIMD (344 registers) ran 48 us, and leaving out pieces
showed that the parts ADD: a lone wave issues at most one instruction of ANY kind per ~4 cycles, so waits, SALU, LDS reads and MFMAs all
queue behind each other.  Two waves per SIMD issue different kinds side by side; hence the diet to 160 + 96 registers: no -mref C
operand (rows that left reference 0 pay a v_sub per score instead, rare), 4-fragment K / V buffers, ring slots as immediates (four
steps per trip).

Row sums, measured and kept as two v_add_f32 per pair (round 5, same box, eval / dropout at [32,4,1000,1000]): one v_pk_add_f32 per pair
(16 fewer vector instructions per phase) 40.9 / 49.4 us against 39.6 / 46.8; the sums on the matrix pipe (a 2-pass 4x4x4 MFMA against a
ones operand per four packed probabilities, no adds at all) 43.3 us: an MFMA holds the wave's in-order issue until the pipe takes it, so
every extra one is a stall for the vector work queued behind it, and the packed add's longer dependent chain costs more than its slot saves.

Structure: a wave owns 64 query rows as two 32-row blocks A and B.
  phase X (X = A, B alternating):  vector port: block X's tile -> p = exp2(s), row sums, bf16 packs;
                                   matrix pipe: the OTHER block's P.V of its previous tile (8 MFMAs), then K.Q^T of its next tile (8).
  step t = phase A(t), phase B(t); one barrier and one counted vmcnt per step; K(t+3) and V(t+2) requested per step by LDS-DMA into
  rings of four 8-KiB slots.
The asm block owns v0-v159 (v64-v95 only after the inputs were read from there), a0-a95, s34-s99.
"""
import os
import sys

NW = 4      # waves per workgroup (64 query rows each): 8 / NW one-KiB LDS-DMA pieces per wave, operand and tile
DMA_AT = "a1"   # where a step's LDS-DMA requests are issued: "a1" = phase A gaps 1.., "b12" = phase B gaps 12.., "tail" = after phase B's last gap
ABL = 0     # timing-only builds (results are garbage): 1 no MFMA, 2 no softmax VALU, 4 no fragment reads, 8 no LDS-DMA, 16 no barrier / vmcnt wait;
            # any value != 0 also removes the range check's branch

# ---- fixed registers -----------------------------------------------------------------------------------------------------------
SA, SB = 0, 32            # score tiles S^T [64 keys x 32 queries] as 2 x 16 accumulator registers each
PA, PB = 64, 80           # packed bf16 probabilities (4 B-operand fragments of 4 registers); the inputs arrive in v64-v95
KF, VF = 96, 112          # fragment buffers: 4 K fragments, 4 V^T fragments (4 registers each)
T = 128                   # 8 rotating exp results
KB, VB = 136, 140         # LDS addresses of the fragment reads in ring slot 0 (K: per k-step; V: [dt][lo/hi])
VOFF = 144                # LDS-DMA source offsets of this lane's two pieces
LA, LB, MA, MB = 146, 147, 148, 149
ACC0, ACC1, TL, TC = 150, 151, 152, 153
SH4, QOFF, WKA, WKB = 154, 155, 156, 158    # SH4 = 4 hh; QOFF doubles as the mask image's lane offset (MOFF) once Q is requested;
MOFF = QOFF                                  # WKA / WKB: the two keep-bit words (32 keys each) of a block's tile, dropout only
NV = 160
OA, OB = 0, 32            # AGPR: O^T accumulators [2 x 16] per block
QA, QB = 64, 80           # AGPR: Q fragments (4 x 4) per block
S_KRS, S_VRS, S_QRS, S_CRS, S_LRS = 36, 40, 44, 48, 52
S_KL, S_NT, S_T, S_KDST, S_VDST = 56, 57, 58, 59, 60
S_KSOFF, S_VSOFF = 61, 62
S_TMP, S_TMP2, S_DSC, S_H128, S_MASKT, S_REM, S_LQ, S_CSIZE, S_CEN, S_SPECIAL = 70, 71, 72, 73, 74, 75, 76, 77, 78, 79
S_STAGE, S_ROWOFF, S_LSE0, S_LK = 90, 91, 92, 93
S_KLP = 94               # (pair) address of this batch element's k_len, or 0
S_MRS, S_M0, S_M1, S_LQP4 = 64, 68, 69, 96     # dropout: the Mk keep-bit image of this (batch, head), offsets of the next tile's two words
DROP = False
S_RET = 80                # return address of the out-of-line pieces
S_SPEC_A, S_SPEC_B, S_RARE_A, S_RARE_B = 82, 84, 86, 88
NINF = "0xff800000"


def v(r, n=1):
    return "v%d" % r if n == 1 else "v[%d:%d]" % (r, r + n - 1)


def a(r, n=1):
    return "a%d" % r if n == 1 else "a[%d:%d]" % (r, r + n - 1)


def s(r, n=1):
    return "s%d" % r if n == 1 else "s[%d:%d]" % (r, r + n - 1)


def rng(base, n):
    return list(range(base, base + n))


class Stream:
    """Instruction list with LDS-queue counting and MFMA-result hazard padding."""

    MFMA_WAIT = 13      # instructions between an 8-pass MFMA and a VALU / memory instruction that touches its result (12 states + 1)

    def __init__(self):
        self.out = []
        self.n = 0                  # wait states issued so far
        self.mfma_w = {}            # VGPR -> wait-state index of the MFMA that last wrote it
        self.lds = []               # pending LDS reads, oldest first: sets of destination VGPRs
        self.nops = 0
        self.counts = {}

    def raw(self, txt, states=1, kind="misc"):
        self.out.append(txt)
        self.n += states
        self.counts[kind] = self.counts.get(kind, 0) + 1

    def comment(self, txt):
        self.out.append("; " + txt)

    def label(self, name):
        self.out.append(name + ":")

    def _need(self, regs):
        last = -1
        for i, dst in enumerate(self.lds):
            if dst & regs:
                last = i
        if last >= 0:
            left = len(self.lds) - 1 - last
            assert left <= 15
            if not (ABL & 64):
                self.raw("s_waitcnt lgkmcnt(%d)" % left, kind="wait")
            self.lds = self.lds[last + 1:]

    def _mfma_pad(self, regs):
        need = 0
        for r in regs:
            if r in self.mfma_w:
                need = max(need, self.mfma_w[r] + self.MFMA_WAIT - self.n)
        while need > 0:
            k = min(need, 8)
            self.raw("s_nop %d" % (k - 1), states=k, kind="nop")
            self.nops += k
            need -= k

    def valu(self, txt, reads=(), writes=(), kind="valu"):
        regs = set(reads) | set(writes)
        self._need(regs)
        self._mfma_pad(regs)
        self.raw(txt, kind=kind)
        for r in writes:
            self.mfma_w.pop(r, None)

    def mfma(self, txt, ab_reads, writes, vgpr_dst):
        self._need(set(ab_reads))
        if ABL & 1:
            return
        self._mfma_pad(set(ab_reads))
        self.raw(txt, kind="mfma")
        if vgpr_dst:
            for r in writes:
                self.mfma_w[r] = self.n - 1

    def lds_read(self, txt, addr, writes):
        if ABL & 4:
            return
        self._need(set(addr))
        self._mfma_pad(set(addr) | set(writes))
        self.raw(txt, kind="lds")
        self.lds.append(set(writes))

    def drain_lds(self):
        self.raw("s_waitcnt lgkmcnt(0)", kind="wait")
        self.lds = []


def blk(X):
    return dict(S=SA, P=PA, L=LA, M=MA, O=OA, Q=QA, SPEC=S_SPEC_A, RARE=S_RARE_A, WK=WKA, MIMM=0) if X == "A" else \
        dict(S=SB, P=PB, L=LB, M=MB, O=OB, Q=QB, SPEC=S_SPEC_B, RARE=S_RARE_B, WK=WKB, MIMM=128)


# ---- pieces of a phase -----------------------------------------------------------------------------------------------------------
def emit_pv_mfma(st, Y, f):
    """P.V MFMA number f (0..7) of block Y: key group g = f >> 1, d half dt = f & 1; its V^T fragment sits in buffer slot f & 3."""
    y = blk(Y)
    g, dt = f >> 1, f & 1
    vf = VF + 4 * (f & 3)
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (a(y["O"] + 16 * dt, 16), v(vf, 4), v(y["P"] + 4 * g, 4), a(y["O"] + 16 * dt, 16)),
            rng(vf, 4) + rng(y["P"] + 4 * g, 4), [], False)


def emit_qk_mfma(st, Y, f):
    """K.Q^T MFMA number f (0..7) of block Y: k-step ks = f >> 1, key half hf = f & 1; K fragment in buffer slot f & 3."""
    y = blk(Y)
    ks, hf = f >> 1, f & 1
    kf = KF + 4 * (f & 3)
    c = "0" if ks == 0 else v(y["S"] + 16 * hf, 16)
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(y["S"] + 16 * hf, 16), v(kf, 4), a(y["Q"] + 4 * ks, 4), c),
            rng(kf, 4), rng(y["S"] + 16 * hf, 16), True)


def emit_k_read(st, f, slot_off, base=KB):
    ks, hf = f >> 1, f & 1
    kf = KF + 4 * (f & 3)
    st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(kf, 4), v(base + ks), slot_off + hf * 4096), [base + ks], rng(kf, 4))


def emit_v_reads(st, f, slot_off, base=VB):
    g, dt = f >> 1, f & 1
    vf = VF + 4 * (f & 3)
    st.lds_read("ds_read_b64_tr_b16 %s, %s offset:%d" % (v(vf, 2), v(base + 2 * dt), slot_off + g * 2048), [base + 2 * dt], rng(vf, 2))
    st.lds_read("ds_read_b64_tr_b16 %s, %s offset:%d" % (v(vf + 2, 2), v(base + 2 * dt + 1), slot_off + g * 2048), [base + 2 * dt + 1], rng(vf + 2, 2))


def emit_exp(st, X, i):
    if ABL & 2:
        return
    x = blk(X)
    st.valu("%s %s, %s" % ("v_mov_b32_e32" if ABL & 128 else "v_exp_f32_e32", v(T + i % 8), v(x["S"] + i)), [x["S"] + i], [T + i % 8], kind="exp")


def emit_sum_pack(st, X, k):
    """row-sum adds and the bf16 pack of exp pair k (scores 2k, 2k+1)."""
    if ABL & 2:
        return
    x = blk(X)
    t0, t1 = T + (2 * k) % 8, T + (2 * k + 1) % 8
    if k == 0:
        st.valu("v_mov_b32_e32 %s, %s" % (v(ACC0), v(t0)), [t0], [ACC0])
        st.valu("v_mov_b32_e32 %s, %s" % (v(ACC1), v(t1)), [t1], [ACC1])
    else:
        st.valu("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(t0)), [ACC0, t0], [ACC0])
        st.valu("v_add_f32_e32 %s, %s, %s" % (v(ACC1), v(ACC1), v(t1)), [ACC1, t1], [ACC1])
    if DROP:         # attention.py:83: the kept probabilities go on to P.V (1 / keep is folded into the final normalisation); the sums above are of all
        for e, t, m in ((2 * k, t0, TC), (2 * k + 1, t1, TL)):
            hf, i = e >> 4, e & 15
            st.valu("v_bfe_i32 %s, %s, %d, 1" % (v(m), v(x["WK"] + hf), 8 * (i >> 2) + (i & 3)), [x["WK"] + hf], [m], kind="drop")
            st.valu("v_and_b32_e32 %s, %s, %s" % (v(t), v(t), v(m)), [t, m], [t], kind="drop")
    st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["P"] + k), v(t0), v(t1)), [t0, t1], [x["P"] + k], kind="cvt")


def emit_mask_loads(st, X):
    """the two keep-bit words of block X's next tile (32 keys each, this lane's query), requested a whole step ahead."""
    x = blk(X)
    st.raw("buffer_load_dword %s, %s, %s, %s offen offset:%d" % (v(x["WK"]), v(MOFF), s(S_MRS, 4), s(S_M0), x["MIMM"]), kind="mload")
    st.raw("buffer_load_dword %s, %s, %s, %s offen offset:%d" % (v(x["WK"] + 1), v(MOFF), s(S_MRS, 4), s(S_M1), x["MIMM"]), kind="mload")


def emit_mask_ready(st, X, outstanding):
    x = blk(X)
    if outstanding is not None:
        st.raw("s_waitcnt vmcnt(%d)" % outstanding, kind="wait")
    st.raw("v_lshrrev_b32_e32 %s, %s, %s" % (v(x["WK"]), v(SH4), v(x["WK"])), kind="drop")          # this lane's keys: bits 8 g + 4 hh + x
    st.raw("v_lshrrev_b32_e32 %s, %s, %s" % (v(x["WK"] + 1), v(SH4), v(x["WK"] + 1)), kind="drop")


def dma_piece(X, gap):
    """the request (0 .. 2 PIECES - 1: K pieces first) issued in this gap, or None."""
    if ABL & 8:
        return None
    n = 2 * (8 // NW)
    if DMA_AT == "a1":
        p = gap - 1 if X == "A" else -1
    elif DMA_AT == "b12":
        p = gap - (16 - n) if X == "B" else -1
    elif DMA_AT == "a8":
        p = gap - 8 if X == "A" else -1
    else:
        p = -1
    return p if 0 <= p < n else None


def emit_dma_m0(st, u, p):
    # (an LDS-DMA load's immediate offset moves its LDS address as well as its source address: M0 takes it back out)
    pieces = 8 // NW
    if p < pieces:
        st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_KDST), ((u + 3) & 3) * 8192 + p * 1024 - (p >> 1) * 2048 + 0x10000), kind="salu")
    else:
        q = p - pieces
        st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_VDST), ((u + 2) & 3) * 8192 + q * 1024 - (q >> 1) * 2048 + 0x10000), kind="salu")


def emit_dma_load(st, p):
    pieces = 8 // NW
    q = p % pieces                    # piece of this wave; pieces q and q + 2 differ by 16 rows = 2048 source bytes (same swizzle)
    st.raw("buffer_load_dwordx4 %s, %s, %s offen offset:%d lds" % (v(VOFF + (q & 1)), s(S_KRS if p < pieces else S_VRS, 4),
                                                                    s(S_KSOFF if p < pieces else S_VSOFF), (q >> 1) * 2048), kind="dma")


def emit_phase(st, X, Y, u, uid):
    """VALU: block X's softmax numerators of tile t.  MFMA: block Y's P.V (tile t-1 for Y = B, t for Y = A) and K.Q^T of its next tile.
    u = t & 3 (ring slots as immediates)."""
    x = blk(X)
    st.comment("---- phase %s: softmax of block %s | P.V and next K.Q^T of block %s" % (uid, X, Y))
    if X == "A":
        v_now, k_slot, v_next = (u + 3) & 3, u, u           # P.V of B(t-1): V(t-1); scores of B(t): K(t); next phase's P.V of A(t): V(t)
    else:
        v_now, k_slot, v_next = u, (u + 1) & 3, u           # P.V of A(t): V(t); scores of A(t+1): K(t+1); next phase's P.V of B(t): V(t)
    # rows that left reference 0, or the ragged last tile: out of line (both rare)
    st.raw("s_cmp_lg_u32 %s, 0" % s(S_SPECIAL), kind="salu")
    st.raw("s_cbranch_scc0 .Lplain_%s_%s" % (uid, "%="), kind="salu")
    st.raw("s_swappc_b64 %s, %s" % (s(S_RET, 2), s(x["SPEC"], 2)), kind="salu")
    st.label(".Lplain_%s_%s" % (uid, "%="))
    if DROP:
        emit_mask_ready(st, X, None if X == "A" else 2 * (8 // NW) + 2)      # (phase A's words: the step's wait in front of the barrier covers them)
    for gap in range(16):
        if gap < 8:
            emit_pv_mfma(st, Y, gap)
        else:
            emit_qk_mfma(st, Y, gap - 8)
        if gap < 4:
            emit_v_reads(st, gap + 4, v_now * 8192)          # V fragments 4..7 of this phase's P.V
        elif gap < 12:
            emit_k_read(st, gap - 4, k_slot * 8192)          # K fragments of this phase's scores, four gaps ahead
        else:
            emit_v_reads(st, gap - 12, v_next * 8192)        # V fragments 0..3 of the NEXT phase's P.V
        dma = dma_piece(X, gap)                              # this step's LDS-DMA requests: K(t+3) -> slot (u+3)&3, V(t+2) -> slot (u+2)&3
        if dma is not None:
            emit_dma_m0(st, u, dma)
        emit_exp(st, X, 2 * gap)
        if dma is not None:
            if ABL & 2:
                st.raw("s_nop 0", kind="nop")
            emit_dma_load(st, dma)
        emit_exp(st, X, 2 * gap + 1)
        if gap >= 1:
            emit_sum_pack(st, X, gap - 1)
    emit_sum_pack(st, X, 15)
    if DMA_AT == "tail" and X == "B" and not (ABL & 8):
        for p in range(2 * (8 // NW)):
            emit_dma_m0(st, u, p)
            st.raw("s_nop 0", kind="nop")
            emit_dma_load(st, p)
    # tail: the lane's partial row sum must stay inside (2^-64, 2^64); anything else takes the whole wave through the re-centring
    if ABL & 32:
        if DROP:
            emit_mask_loads(st, X)
        return
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(ACC1)), [ACC0, ACC1], [ACC0])
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(TL), v(x["L"]), v(ACC0)), [x["L"], ACC0], [TL])
    st.valu("v_add_u32_e32 %s, 0xe0800000, %s" % (v(TC), v(TL)), [TL], [TC])
    st.valu("v_cmp_le_u32_e32 vcc, 0x40000000, %s" % v(TC), [TC], [])
    if not ABL:
        st.raw("s_cbranch_vccz .Lfine_%s_%s" % (uid, "%="), kind="salu")
        st.raw("s_swappc_b64 %s, %s" % (s(S_RET, 2), s(x["RARE"], 2)), kind="salu")
        st.label(".Lfine_%s_%s" % (uid, "%="))
    st.valu("v_mov_b32_e32 %s, %s" % (v(x["L"]), v(TL)), [TL], [x["L"]])
    if DROP:
        emit_mask_loads(st, X)


def emit_special(st, X):
    """out of line, at a phase's start: s -= mref for a wave whose rows left reference 0; the ragged last tile's keys past k_len -> -inf."""
    x = blk(X)
    st.label(".Lspecial_%s_%s" % (X, "%="))
    st.raw("s_nop 7")
    st.raw("s_nop 7")                 # (the scores' last MFMA is at least 12 wait states back on every path; this is not a hot path)
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_CEN))
    st.raw("s_cbranch_scc1 .Lnocen_%s_%s" % (X, "%="))
    for i in range(32):
        st.raw("v_sub_f32_e32 %s, %s, %s" % (v(x["S"] + i), v(x["S"] + i), v(x["M"])))
    st.label(".Lnocen_%s_%s" % (X, "%="))
    st.raw("s_cmp_eq_u32 %s, %s" % (s(S_T), s(S_MASKT)))
    st.raw("s_cbranch_scc0 .Lnomask_%s_%s" % (X, "%="))
    st.raw("v_sub_u32_e32 %s, %s, %s" % (v(TC), s(S_REM), v(SH4)))            # keys of this tile that exist, minus 4 hh
    st.raw("v_mov_b32_e32 %s, %s" % (v(T + 7), NINF))
    for hf in range(2):
        for i in range(16):
            key = 32 * hf + (i & 3) + 8 * (i >> 2)
            st.raw("v_cmp_lt_i32_e32 vcc, %d, %s" % (key, v(TC)))
            st.raw("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(x["S"] + 16 * hf + i), v(T + 7), v(x["S"] + 16 * hf + i)))
    st.label(".Lnomask_%s_%s" % (X, "%="))
    st.raw("s_setpc_b64 %s" % s(S_RET, 2))


def emit_xaddr(st, reg):
    """ds_bpermute address of the lane that holds the other half of this lane's query row: ((lane ^ 32) << 2)."""
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(reg))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(reg), v(reg)))
    st.raw("v_xor_b32_e32 %s, 32, %s" % (v(reg), v(reg)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(reg), v(reg)))


def emit_rare(st, X):
    """out of line, at a phase's tail: re-centre block X (a change of reference, exact up to rounding) and redo its tile's numerators;
    returns with TL = the lane's new partial sum."""
    x = blk(X)
    R = KF            # scratch: the K fragment buffer is idle at a phase's tail (16 registers), and so are the exp results
    S_ = x["S"]
    st.label(".Lrare_%s_%s" % (X, "%="))
    st.raw("v_max3_f32 %s, %s, %s, %s" % (v(R), v(S_), v(S_ + 1), v(S_ + 2)))
    for i in range(3, 31, 2):
        st.raw("v_max3_f32 %s, %s, %s, %s" % (v(R), v(R), v(S_ + i), v(S_ + i + 1)))
    st.raw("v_max_f32_e32 %s, %s, %s" % (v(R), v(R), v(S_ + 31)))
    st.raw("s_waitcnt lgkmcnt(0)")
    emit_xaddr(st, R + 10)
    st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R + 1), v(R + 10), v(R)))
    st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R + 2), v(R + 10), v(x["L"])))
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("v_max_f32_e32 %s, %s, %s" % (v(R), v(R), v(R + 1)))                       # the row's maximum (relative to mref)
    st.raw("v_add_f32_e32 %s, %s, %s" % (v(R + 2), v(R + 2), v(x["L"])))              # the row's sum so far
    st.raw("v_log_f32_e32 %s, %s" % (v(R + 3), v(R + 2)))                               # log2; -inf for an empty row
    st.raw("s_nop 1")
    st.raw("v_max_f32_e32 %s, %s, %s" % (v(R + 4), v(R), v(R + 3)))                   # delta
    st.raw("v_mov_b32_e32 %s, %s" % (v(R + 7), NINF))
    st.raw("v_cmp_lt_f32_e32 vcc, %s, %s" % (v(R + 7), v(R + 4)))                      # false for -inf and NaN
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(R + 4), v(R + 4)))
    st.raw("v_exp_f32_e64 %s, -%s" % (v(R + 5), v(R + 4)))                              # alpha
    st.raw("v_cmp_lt_f32_e32 vcc, 0, %s" % v(R + 2))                                    # row not empty
    st.raw("s_nop 1")
    st.raw("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(R + 5), v(R + 5)))                  # alpha (0 for an empty row: its O and l are 0)
    st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 6), v(R + 2), v(R + 5)))
    st.raw("v_mul_f32_e32 %s, 0.5, %s" % (v(R + 6), v(R + 6)))                          # the row's rescaled sum, shared evenly by its two lanes
    st.raw("v_add_f32_e32 %s, %s, %s" % (v(x["M"]), v(x["M"]), v(R + 4)))
    st.raw("s_nop 7")
    st.raw("s_nop 7")
    for i in range(32):
        st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + 8), a(x["O"] + i)))
        st.raw("s_nop 1")
        st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 8), v(R + 8), v(R + 5)))
        st.raw("s_nop 1")
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(x["O"] + i), v(R + 8)))
    for i in range(32):
        st.raw("v_sub_f32_e32 %s, %s, %s" % (v(S_ + i), v(S_ + i), v(R + 4)))
    for k in range(16):
        st.raw("v_exp_f32_e32 %s, %s" % (v(R + 8), v(S_ + 2 * k)))
        st.raw("v_exp_f32_e32 %s, %s" % (v(R + 9), v(S_ + 2 * k + 1)))
        st.raw("s_nop 1")
        if k == 0:
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(R + 8), v(R + 9)))
        else:
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(R + 8)))
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(R + 9)))
        if DROP:
            for e, t in ((2 * k, R + 8), (2 * k + 1, R + 9)):
                hf, i = e >> 4, e & 15
                st.raw("v_bfe_i32 %s, %s, %d, 1" % (v(R + 10), v(x["WK"] + hf), 8 * (i >> 2) + (i & 3)))
                st.raw("v_and_b32_e32 %s, %s, %s" % (v(t), v(t), v(R + 10)))
        st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["P"] + k), v(R + 8), v(R + 9)))
    st.raw("v_add_f32_e32 %s, %s, %s" % (v(TL), v(R + 6), v(ACC0)))
    st.raw("s_mov_b32 %s, 1" % s(S_CEN))
    st.raw("s_mov_b32 %s, 1" % s(S_SPECIAL))
    st.raw("s_setpc_b64 %s" % s(S_RET, 2))


def emit_dma_now(st, rsrc, soff, dst_sgpr, imm, piece):
    st.raw("s_add_u32 m0, %s, 0x%x" % (s(dst_sgpr), imm - (piece >> 1) * 2048 + 0x10000))
    st.raw("s_nop 0")
    st.raw("buffer_load_dwordx4 %s, %s, %s offen offset:%d lds" % (v(VOFF + (piece & 1)), s(rsrc, 4), s(soff), (piece >> 1) * 2048))


def build(drop):
    st = Stream()
    U = "%="
    st.comment("==== attention forward v4 (%s): generated by tools/gen_attn_fwd4.py - do not edit" % ("train: dropout" if drop else "eval"))
    st.comment("---- inputs (they sit in v64-v95 / low SGPRs) into the block's own registers")
    st.raw("v_mov_b32_e32 %s, %%[qoff]" % v(QOFF))
    st.raw("s_mov_b64 %s, %%[qb]" % s(S_QRS, 2))
    st.raw("s_and_b32 %s, %s, 0xffff" % (s(S_QRS + 1), s(S_QRS + 1)))
    st.raw("s_mov_b32 %s, 0x00020000" % s(S_QRS + 3))
    st.raw("s_mov_b32 %s, %%[lq]" % s(S_LQ))
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_QRS + 2), s(S_LQ)))
    st.comment("---- first of all: this batch element's k_len and the Q fragments of both blocks (block B: 32 rows = 4096 bytes further)")
    st.raw("s_mov_b64 %s, %%[klp]" % s(S_KLP, 2))
    st.raw("s_mov_b32 %s, %%[lk]" % s(S_LK))
    st.raw("s_mov_b32 %s, %s" % (s(S_KL), s(S_LK)))
    st.raw("s_cmp_eq_u64 %s, 0" % s(S_KLP, 2))
    st.raw("s_cbranch_scc1 .Lnoklen_%=")
    st.raw("s_load_dword %s, %s, 0x0" % (s(S_KL), s(S_KLP, 2)))
    st.label(".Lnoklen_%=")
    st.raw("v_add_u32_e32 %s, 0x1000, %s" % (v(TL), v(QOFF)))
    st.raw("s_nop 2")
    for ks in range(4):
        st.raw(("s_nop 0" if ABL & 512 else "buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SA + 4 * ks, 4), v(QOFF), s(S_QRS, 4), 32 * ks)))
    for ks in range(4):
        st.raw(("s_nop 0" if ABL & 512 else "buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(SA + 16 + 4 * ks, 4), v(TL), s(S_QRS, 4), 32 * ks)))
    for name, reg in (("voff0", VOFF), ("voff1", VOFF + 1), ("kofs0", KB), ("kofs1", KB + 1), ("kofs2", KB + 2), ("kofs3", KB + 3),
                      ("vofs0", VB), ("vofs1", VB + 1), ("vofs2", VB + 2), ("vofs3", VB + 3), ("sh4", SH4)):
        st.raw("v_mov_b32_e32 %s, %%[%s]" % (v(reg), name))
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(TC))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(TC), v(TC)))
    st.raw("v_lshlrev_b32_e32 %s, 4, %s" % (v(TC), v(TC)))                # lane * 16: this lane's bytes of a 1-KiB zero-fill row
    for name, reg in (("kb", S_KRS), ("vb", S_VRS), ("cb", S_CRS), ("lb", S_LRS)):
        st.raw("s_mov_b64 %s, %%[%s]" % (s(reg, 2), name))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(reg + 1), s(reg + 1)))
        st.raw("s_mov_b32 %s, 0x00020000" % s(reg + 3))
    for name, reg in (("kdst", S_KDST), ("dsc", S_DSC), ("h128", S_H128), ("csize", S_CSIZE), ("stage", S_STAGE), ("lse0", S_LSE0)):
        st.raw("s_mov_b32 %s, %%[%s]" % (s(reg), name))
    st.raw("s_lshr_b32 %s, %s, 2" % (s(S_TMP), s(S_LSE0)))
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_ROWOFF), s(S_TMP), s(S_H128)))       # ctx byte offset of this wave's first row
    st.raw("s_lshl_b32 %s, %s, 2" % (s(S_LRS + 2), s(S_LQ)))
    st.raw("s_cmp_eq_u64 %s, 0" % s(S_LRS, 2))                                # no lse wanted: an empty buffer drops the stores
    st.raw("s_cselect_b32 %s, 0, %s" % (s(S_LRS + 2), s(S_LRS + 2)))
    if DROP:
        st.raw("s_mov_b64 %s, %%[mb]" % s(S_MRS, 2))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(S_MRS + 1), s(S_MRS + 1)))
        st.raw("s_mov_b32 %s, %%[msz]" % s(S_MRS + 2))
        st.raw("s_mov_b32 %s, 0x00020000" % s(S_MRS + 3))
        st.raw("s_mov_b32 %s, %%[lqp4]" % s(S_LQP4))
        st.raw("v_lshrrev_b32_e32 %s, 7, %s" % (v(MOFF), v(QOFF)))             # (the Q loads are out: QOFF becomes 4 * query row)
        st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(MOFF), v(MOFF)))
    st.raw("s_mov_b32 %s, %s" % (s(S_CRS + 2), s(S_CSIZE)))
    st.raw("s_sub_u32 %s, %s, 0x10000" % (s(S_KDST), s(S_KDST)))          # (biased: the M0 sums add it back; keeps every immediate positive)
    st.raw("s_add_u32 %s, %s, 0x8000" % (s(S_VDST), s(S_KDST)))
    st.raw("s_waitcnt lgkmcnt(0)")                                        # k_len
    st.raw("s_min_i32 %s, %s, %s" % (s(S_KL), s(S_KL), s(S_LK)))
    st.raw("s_max_i32 %s, %s, 0" % (s(S_KL), s(S_KL)))
    st.raw("s_add_u32 %s, %s, 63" % (s(S_NT), s(S_KL)))
    st.raw("s_lshr_b32 %s, %s, 6" % (s(S_NT), s(S_NT)))
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_KRS + 2), s(S_KL)))             # K and V: k_len rows of 128 bytes (anything past them reads as zeros)
    st.raw("s_mov_b32 %s, %s" % (s(S_VRS + 2), s(S_KRS + 2)))
    st.raw("s_and_b32 %s, %s, 63" % (s(S_TMP), s(S_KL)))                  # the step that holds a ragged last tile (none: -1)
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_MASKT), s(S_NT)))
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_TMP))
    st.raw("s_cselect_b32 %s, -1, %s" % (s(S_MASKT), s(S_MASKT)))
    st.raw("s_mov_b32 %s, 0" % s(S_CEN))
    st.raw("s_mov_b32 %s, 0" % s(S_T))
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_MASKT))
    st.raw("s_cselect_b32 %s, 1, 0" % s(S_SPECIAL))
    st.raw("s_mov_b32 %s, %s" % (s(S_REM), s(S_KL)))
    # addresses of the out-of-line pieces
    st.raw("s_getpc_b64 %s" % s(S_TMP, 2))
    st.label(".Lhere_" + U)
    for reg, lab in ((S_SPEC_A, ".Lspecial_A_"), (S_SPEC_B, ".Lspecial_B_"), (S_RARE_A, ".Lrare_A_"), (S_RARE_B, ".Lrare_B_")):
        st.raw("s_add_u32 %s, %s, %s%s-.Lhere_%s" % (s(reg), s(S_TMP), lab, U, U))
        st.raw("s_addc_u32 %s, %s, 0" % (s(reg + 1), s(S_TMP2)))
    st.comment("---- K0 V0 K1 V1 K2 by LDS-DMA")
    st.raw("s_nop 4")
    for tile, which in ((0, "K"), (0, "V"), (1, "K"), (1, "V"), (2, "K")):
        st.raw("s_mov_b32 %s, 0x%x" % (s(S_TMP), tile * 8192))
        for piece in range(8 // NW if not (ABL & 1024) else 0):
            emit_dma_now(st, S_KRS if which == "K" else S_VRS, S_TMP, S_KDST if which == "K" else S_VDST, tile * 8192 + piece * 1024, piece)
    if DROP:
        st.raw("s_mov_b32 %s, 0" % s(S_M0))
        st.raw("s_mov_b32 %s, %s" % (s(S_M1), s(S_LQP4)))
        st.raw("s_nop 2")
        emit_mask_loads(st, "A")
        emit_mask_loads(st, "B")
        st.raw("s_lshl_b32 %s, %s, 1" % (s(S_M0), s(S_LQP4)))             # next: tile 1's words (key groups 2 and 3)
        st.raw("s_add_u32 %s, %s, %s" % (s(S_M1), s(S_M0), s(S_LQP4)))
    st.raw("s_mov_b32 %s, 0x%x" % (s(S_KSOFF), 3 * 8192))             # the requests of step 0: K(3), V(2)
    st.raw("s_mov_b32 %s, 0x%x" % (s(S_VSOFF), 2 * 8192))
    st.comment("---- state: O = 0, l = 0, mref = 0, P of block B = 0, V ring slot 3 = 0 (the first phase multiplies it by that P)")
    for i in range(64):
        st.raw("v_accvgpr_write_b32 %s, 0" % a(i))
    for r in rng(PB, 16) + [LA, LB, MA, MB]:
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    for r in range(4):
        st.raw("v_mov_b32_e32 %s, 0" % v(T + r))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(TC), s(S_VDST), v(TC)))
    st.raw("v_add_u32_e32 %s, 0x10000, %s" % (v(TC), v(TC)))
    for piece in range(8 // NW):
        st.raw("ds_write_b128 %s, %s offset:%d" % (v(TC), v(T, 4), 3 * 8192 + piece * 1024))
    st.comment("---- Q into the accumulator file (B operands of every K.Q^T)")
    st.raw("s_waitcnt vmcnt(%d)" % ((5 * (8 // NW) if not (ABL & 1536) else 0) + (4 if DROP else 0)))
    for i in range(32):
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(QA + i), v(SA + i)))
    st.raw("s_waitcnt vmcnt(%d) lgkmcnt(0)" % ((4 * (8 // NW) if not (ABL & 1536) else 0) + (4 if DROP else 0)))        # K0 landed (V0 K1 V1 K2 may be in flight); the zero fill is in LDS
    st.raw("s_barrier")
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_NT))
    st.raw("s_cbranch_scc1 .Lfinal_" + U)
    st.comment("---- K.Q^T of block A's tile 0; V fragments 0..3 of the (zero) tile -1")
    st.raw("s_nop 7", states=8)
    for half in range(2):
        for f in range(4 * half, 4 * half + 4):
            emit_k_read(st, f, 0)
        for f in range(4 * half, 4 * half + 4):
            emit_qk_mfma(st, "A", f)
    for f in range(4):
        emit_v_reads(st, f, 3 * 8192)
    # ---- the loop: four steps per trip (ring slots as immediates); two passes, the first only to learn the state at the back edge ----------
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
            st.comment("==== step t, t & 3 == %d" % u)
            if not (ABL & 16):
                st.raw("s_waitcnt vmcnt(%d)" % (2 if DROP else 2 * (8 // NW)), kind="wait")      # (dropout: all but block B's two mask words) K(t+1) and V(t) have landed: everything but the previous step's four requests
                st.raw("s_barrier", kind="salu")
            emit_phase(st, "A", "B", u, "a%d" % u)
            emit_phase(st, "B", "A", u, "b%d" % u)
            st.raw("s_add_u32 %s, %s, 1" % (s(S_T), s(S_T)), kind="salu")
            st.raw("s_add_u32 %s, %s, 0x2000" % (s(S_KSOFF), s(S_KSOFF)), kind="salu")
            st.raw("s_add_u32 %s, %s, 0x2000" % (s(S_VSOFF), s(S_VSOFF)), kind="salu")
            st.raw("s_sub_u32 %s, %s, 64" % (s(S_REM), s(S_REM)), kind="salu")
            if DROP:
                st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M0), s(S_LQP4), s(S_M0)), kind="salu")
                st.raw("s_lshl1_add_u32 %s, %s, %s" % (s(S_M1), s(S_LQP4), s(S_M1)), kind="salu")
            st.raw("s_cmp_eq_u32 %s, %s" % (s(S_T), s(S_MASKT)), kind="salu")
            st.raw("s_cselect_b32 %s, 1, %s" % (s(S_SPECIAL), s(S_CEN)), kind="salu")
            st.raw("s_cmp_lt_u32 %s, %s" % (s(S_T), s(S_NT)), kind="salu")
            if u < 3:
                st.raw("s_cbranch_scc0 .Ldone_" + U, kind="salu")
            else:
                st.raw("s_cbranch_scc1 .Ltrip_" + U, kind="salu")
    body_counts = dict(st.counts)
    body_nops = st.nops
    # ---- after the last step (t = nt now): block B's last P.V; its V fragments 4..7 are still to be read, from slot (nt - 1) & 3 ------------
    st.label(".Ldone_" + U)
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_TMP), s(S_NT)))
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_TMP)))
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_TMP), s(S_TMP)))
    for i in range(4):
        st.raw("v_add_u32_e32 %s, %s, %s" % (v(T + i), s(S_TMP), v(VB + i)))
    for f in range(4):
        emit_pv_mfma(st, "B", f)
    st.raw("s_nop 3", states=4)
    for f in range(4, 8):
        emit_v_reads(st, f, 0, base=T)
    for f in range(4, 8):
        emit_pv_mfma(st, "B", f)
    st.label(".Lfinal_" + U)
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")             # (out-of-range requests of the last steps: nothing may land after the LDS is released)
    st.raw("s_nop 7", states=8)
    st.raw("s_nop 7", states=8)
    st.comment("---- epilogue: O / l -> bf16, through LDS (rows of 144 bytes: 128 + a pad that spreads the banks) to whole-row stores; lse = mref + log2(l)")
    # every wave of the workgroup is past its last fragment read and its last LDS-DMA request has landed: the rings are free
    st.raw("s_barrier")
    LN, RR, WA, RA, SO, XA = T, T + 1, T + 2, T + 3, T + 4, T + 5
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(LN))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(LN), v(LN)))
    st.raw("v_and_b32_e32 %s, 31, %s" % (v(RR), v(LN)))                                  # r: the lane's query row inside a block
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(WA), v(LN)))                                # hh
    st.raw("v_lshlrev_b32_e32 %s, 3, %s" % (v(WA), v(WA)))
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(WA), v(RR), v(WA)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(WA), v(RR), v(WA)))                      # + 144 r
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(WA), s(S_STAGE), v(WA)))                      # write address: row r, byte 8 hh (+ 64 dt + 16 g; block B 32 rows on)
    st.raw("v_lshrrev_b32_e32 %s, 3, %s" % (v(RA), v(LN)))                                # row of the read-back: lane >> 3 (+ 8 per instruction)
    st.raw("v_and_b32_e32 %s, 7, %s" % (v(SO), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 4, %s" % (v(SO), v(SO)))                                # 16-byte chunk lane & 7
    st.raw("v_mul_lo_u32 %s, %s, %s" % (v(XA), v(RA), s(S_H128)))
    st.raw("v_lshl_add_u32 %s, %s, 7, %s" % (v(SO + 2), v(RA), v(SO)))
    st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(RA), v(RA), v(SO + 2)))                  # 144 (lane >> 3) + chunk
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(RA), s(S_STAGE), v(RA)))                      # read address
    st.raw("v_add3_u32 %s, %s, %s, %s" % (v(SO), v(XA), v(SO), s(S_ROWOFF)))             # ctx byte offset of (row, chunk)
    st.raw("v_xor_b32_e32 %s, 32, %s" % (v(XA), v(LN)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(XA), v(XA)))
    R = SA            # the score registers are free now
    for bi, X in enumerate(("A", "B")):
        x = blk(X)
        st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R), v(XA), v(x["L"])))
        st.raw("s_waitcnt lgkmcnt(0)")
        st.raw("v_add_f32_e32 %s, %s, %s" % (v(R), v(R), v(x["L"])))
        st.raw("v_rcp_f32_e32 %s, %s" % (v(R + 1), v(R)))
        st.raw("v_log_f32_e32 %s, %s" % (v(R + 2), v(R)))
        st.raw("s_nop 1", states=2)
        if drop:
            st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 1), s(S_DSC), v(R + 1)))
        st.raw("v_add_f32_e32 %s, %s, %s" % (v(R + 2), v(R + 2), v(x["M"])))
        for dt in range(2):
            for g in range(4):
                for j in range(4):
                    st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + 4 + j), a(x["O"] + 16 * dt + 4 * g + j)))
                st.raw("s_nop 0")
                for j in range(4):
                    st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 4 + j), v(R + 4 + j), v(R + 1)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 8), v(R + 4), v(R + 5)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 9), v(R + 6), v(R + 7)))
                st.raw("ds_write_b64 %s, %s offset:%d" % (v(WA), v(R + 8, 2), bi * 32 * 144 + 64 * dt + 16 * g))
        # lse of the block's 32 rows: lanes 0..31 hold one row each
        st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(R + 3), v(RR)))
        st.raw("v_add_u32_e32 %s, %s, %s" % (v(R + 3), s(S_LSE0), v(R + 3)))
        st.raw("s_mov_b32 exec_hi, 0")
        if not (ABL & 256):
            st.raw("buffer_store_dword %s, %s, %s, 0 offen offset:%d" % (v(R + 2), v(R + 3), s(S_LRS, 4), 128 * bi))
        st.raw("s_mov_b32 exec_hi, -1")
        R = SB
    st.raw("s_waitcnt lgkmcnt(0)")                     # the wave reads back only what it wrote itself: no barrier
    for i in range(8):
        st.raw("ds_read_b128 %s, %s offset:%d" % (v(SA + 4 * i, 4), v(RA), i * 8 * 144))
    st.raw("s_mov_b32 %s, 0" % s(S_TMP))
    st.raw("s_lshl_b32 %s, %s, 3" % (s(S_TMP2), s(S_H128)))                # 8 rows of ctx
    for i in range(8):
        st.raw("s_waitcnt lgkmcnt(%d)" % (7 - i))
        if not (ABL & 256):
            st.raw("buffer_store_dwordx4 %s, %s, %s, %s offen" % (v(SA + 4 * i, 4), v(SO), s(S_CRS, 4), s(S_TMP)))
        st.raw("s_add_u32 %s, %s, %s" % (s(S_TMP), s(S_TMP), s(S_TMP2)))
    st.raw("s_endpgm")
    for X in ("A", "B"):
        emit_special(st, X)
        emit_rare(st, X)
    return st, body_counts, body_nops


def main():
    global ABL, NW, DMA_AT, DROP
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "end-to-end_asr_pytorch_amd", "csrc", "attention_fwd4_asm.inc")
    args = sys.argv[1:]
    while args and args[0].startswith("--"):          # tools/abl_attn4.sh: python tools/gen_attn_fwd4.py --abl 3 --dma tail --out /tmp/x.inc
        if args[0] == "--abl":
            ABL = int(args[1])
        elif args[0] == "--dma":
            DMA_AT = args[1]
        elif args[0] == "--out":
            out = args[1]
        args = args[2:]
    with open(out, "w") as f:
        f.write("// generated by tools/gen_attn_fwd4.py - do not edit (edit the generator and run it again)\n")
        for nw in (2, 4):
            NW = nw
            for drop in (False, True):
                DROP = drop
                st, counts, nops = build(drop)
                f.write("#define ATTN4_ASM_%s_NW%d \\\n" % ("TRAIN" if drop else "EVAL", nw))
                for line in st.out:
                    f.write('    "%s\\n" \\\n' % line.replace("\\", "\\\\").replace('"', '\\"'))
                f.write('    ""\n')
                sys.stderr.write("NW %d %s: %d lines; per trip of 4 steps: %s; s_nop states padded in the loop: %d\n" %
                                 (nw, "train" if drop else "eval", len(st.out), counts, nops))
        regs = ["v%d" % i for i in list(range(64)) + list(range(96, NV))] + ["a%d" % i for i in range(96)] + ["s%d" % i for i in range(34, 100)] + ["vcc", "memory"]
        f.write("#define ATTN4_ASM_CLOBBERS %s\n" % ", ".join('"%s"' % r for r in regs))


if __name__ == "__main__":
    main()
