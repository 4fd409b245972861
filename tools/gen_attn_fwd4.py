#!/usr/bin/env python3
"""Generator of the encoder self-attention forward's instruction stream (csrc/attention_fwd4_asm.inc, included by attention_fwd4.hip).

Why generated assembly: at d_k = 64 the loop is bound by vector-instruction ISSUE (32 v_exp_f32 + 32 v_add_f32 + 16 v_cvt_pk per
16 MFMAs on one SIMD), so every MFMA gap has to carry its share of the softmax, its LDS fragment reads and its counted waits - hipcc
bunches the MFMAs and the VALU block instead (80 us against v3's 54), with sched_group_barrier spills or moves the accumulators to
AGPRs and pays a v_accvgpr_read per score.  This script assigns every instruction of a phase to an MFMA gap from a table, counts the
LDS queue for the s_waitcnt lgkmcnt(N) in front of each MFMA, and checks the MFMA -> VALU read distances (12 wait states, 8-pass XDL).

Structure (see attention_fwd4.hip's header): a wave owns 64 query rows as two 32-row blocks A and B.
  phase X (X = A, B alternating):  vector port: block X's tile -> p = exp2(s), row sums, bf16 packs;
                                   matrix pipe: the OTHER block's P.V of its previous tile (8 MFMAs), then its K.Q^T of its next tile (8).
  step t = phase A(t), phase B(t); one barrier and one counted vmcnt per step; K(t+3) and V(t+2) requested per step by LDS-DMA into
  rings of four 8-KiB slots.
Registers are fixed (the asm block owns v0-v231, a0-a95, s34-s99); inputs arrive in compiler-chosen registers outside those ranges.
"""
import os
import sys

# ---- fixed registers -----------------------------------------------------------------------------------------------------------
SA, SB = 0, 32            # score tiles S^T [64 keys x 32 queries] as 2 x 16 accumulator registers each
PA, PB = 64, 80           # packed bf16 probabilities (4 B-operand fragments of 4 registers)
NA, NB = 96, 112          # -mref broadcast over 16 registers: the C operand of a tile's first K.Q^T MFMA
KF, VF = 128, 160         # fragment buffers: 8 K fragments, 8 V^T fragments (4 registers each)
T = 192                   # 8 rotating exp results
KB, VB = 200, 204         # LDS addresses of the fragment reads in ring slot 0 (K: per k-step; V: [dt][lo/hi])
KAD, VAD = 208, 212       # the same for the slots of the running phase
VOFF = 216                # LDS-DMA source offsets of this lane's two pieces
LA, LB, MA, MB = 218, 219, 220, 221
ACC0, ACC1, TL, TC = 222, 223, 224, 225
THR, NINF, XADDR, OOFF, LSEOFF, QOFF = 226, 227, 228, 229, 230, 231
OA, OB = 0, 32            # AGPR: O^T accumulators [2 x 16] per block
QA, QB = 64, 80           # AGPR: Q fragments (4 x 4) per block
S_KRS, S_VRS, S_QRS, S_CRS, S_LRS = 36, 40, 44, 48, 52
S_KL, S_NT, S_T, S_KDST, S_VDST = 56, 57, 58, 59, 60
S_KSOFF, S_VSOFF, S_D0, S_D1, S_D2, S_D3 = 61, 62, 63, 64, 65, 66
S_KCUR, S_KNXT, S_VCUR, S_TMP, S_TMP2, S_DSC, S_HROW, S_MASKT, S_REM = 67, 68, 69, 70, 71, 72, 73, 74, 75
S_LQ, S_CSIZE = 76, 77


def v(r, n=1):
    return "v%d" % r if n == 1 else "v[%d:%d]" % (r, r + n - 1)


def a(r, n=1):
    return "a%d" % r if n == 1 else "a[%d:%d]" % (r, r + n - 1)


def s(r, n=1):
    return "s%d" % r if n == 1 else "s[%d:%d]" % (r, r + n - 1)


ABL = 0     # timing-only builds (results are garbage): 1 no MFMA, 2 no softmax VALU, 4 no fragment reads, 8 no LDS-DMA, 16 no barrier / vmcnt wait,
            # 32 no MFMA-result padding check (unused); any value != 0 also removes the range check's branch


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
        # wait for pending LDS reads that write any of regs
        last = -1
        for i, dst in enumerate(self.lds):
            if dst & regs:
                last = i
        if last >= 0:
            left = len(self.lds) - 1 - last
            assert left <= 15
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

    def mfma(self, txt, ab_reads, c_reads, writes, vgpr_dst):
        if ABL & 1:
            self._need(set(ab_reads) | set(c_reads))
            return
        self._need(set(ab_reads) | set(c_reads))
        self._mfma_pad(set(ab_reads))                       # an MFMA result as the next MFMA's A / B operand (not used; C chains are free)
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


# ---- pieces ----------------------------------------------------------------------------------------------------------------------
def blk(X):
    return dict(S=SA, P=PA, N=NA, L=LA, M=MA, O=OA, Q=QA) if X == "A" else dict(S=SB, P=PB, N=NB, L=LB, M=MB, O=OB, Q=QB)


def rng(base, n):
    return list(range(base, base + n))


def emit_pv_mfma(st, Y, f):
    """P.V MFMA number f (0..7) of block Y: key group g = f >> 1, d half dt = f & 1."""
    y = blk(Y)
    g, dt = f >> 1, f & 1
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (a(y["O"] + 16 * dt, 16), v(VF + 4 * f, 4), v(y["P"] + 4 * g, 4), a(y["O"] + 16 * dt, 16)),
            rng(VF + 4 * f, 4) + rng(y["P"] + 4 * g, 4), [], [], False)


def emit_qk_mfma(st, Y, f):
    """K.Q^T MFMA number f (0..7) of block Y: k-step ks = f >> 1, key half hf = f & 1."""
    y = blk(Y)
    ks, hf = f >> 1, f & 1
    c = v(y["N"], 16) if ks == 0 else v(y["S"] + 16 * hf, 16)
    creads = rng(y["N"], 16) if ks == 0 else []
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(y["S"] + 16 * hf, 16), v(KF + 4 * f, 4), a(y["Q"] + 4 * ks, 4), c),
            rng(KF + 4 * f, 4), creads, rng(y["S"] + 16 * hf, 16), True)


def emit_k_read(st, f):
    ks, hf = f >> 1, f & 1
    st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(KF + 4 * f, 4), v(KAD + ks), hf * 4096), [KAD + ks], rng(KF + 4 * f, 4))


def emit_v_reads(st, f):
    g, dt = f >> 1, f & 1
    st.lds_read("ds_read_b64_tr_b16 %s, %s offset:%d" % (v(VF + 4 * f, 2), v(VAD + 2 * dt), g * 2048), [VAD + 2 * dt], rng(VF + 4 * f, 2))
    st.lds_read("ds_read_b64_tr_b16 %s, %s offset:%d" % (v(VF + 4 * f + 2, 2), v(VAD + 2 * dt + 1), g * 2048), [VAD + 2 * dt + 1], rng(VF + 4 * f + 2, 2))


def emit_exp(st, X, i):
    if ABL & 2:
        return
    x = blk(X)
    st.valu("v_exp_f32_e32 %s, %s" % (v(T + i % 8), v(x["S"] + i)), [x["S"] + i], [T + i % 8], kind="exp")


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
    st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["P"] + k), v(t0), v(t1)), [t0, t1], [x["P"] + k], kind="cvt")


def emit_dma(st, rsrc, soff, dst, piece, loop=False):
    if loop and (ABL & 8):
        return
    st.raw("s_mov_b32 m0, %s" % s(dst), kind="salu")
    st.raw("s_nop 0", kind="nop")
    st.raw("buffer_load_dwordx4 %s, %s, %s offen lds" % (v(VOFF + piece), s(rsrc, 4), s(soff)), kind="dma")


def emit_mask(st, X, uid):
    """the ragged last tile: keys past k_len (their K rows were read as zeros) get score -inf.  scc = this step holds that tile."""
    x = blk(X)
    st.raw("s_cbranch_scc0 .Lnomask_%s_%s" % (uid, "%="), kind="salu")
    st.valu("v_add_u32_e32 %s, %s, %s" % (v(TC), s(S_REM), v(THR)), [THR], [TC])      # keys of this tile that exist, minus 4 hh
    for hf in range(2):
        for i in range(16):
            key = 32 * hf + (i & 3) + 8 * (i >> 2)
            st.valu("v_cmp_lt_i32_e32 vcc, %d, %s" % (key, v(TC)), [TC], [])
            st.valu("v_cndmask_b32_e32 %s, %s, %s, vcc" % (v(x["S"] + 16 * hf + i), v(NINF), v(x["S"] + 16 * hf + i)), [NINF, x["S"] + 16 * hf + i], [x["S"] + 16 * hf + i])
    st.label(".Lnomask_%s_%s" % (uid, "%="))


def emit_phase(st, X, Y, uid, first_dma=False):
    """VALU: block X's softmax numerators.  MFMA: block Y's P.V (fragments already in VF) and K.Q^T of its next tile."""
    x = blk(X)
    st.comment("---- phase %s: softmax of block %s | P.V and next K.Q^T of block %s" % (uid, X, Y))
    # fragment addresses of this phase: K tile of Y's next scores; V tile of the next phase's P.V
    koff = S_KCUR if X == "A" else S_KNXT
    for i in range(4):
        st.valu("v_add_u32_e32 %s, %s, %s" % (v(KAD + i), s(koff), v(KB + i)), [KB + i], [KAD + i])
    if X == "A":
        for i in range(4):
            st.valu("v_add_u32_e32 %s, %s, %s" % (v(VAD + i), s(S_VCUR), v(VB + i)), [VB + i], [VAD + i])
    st.raw("s_cmp_eq_u32 %s, %s" % (s(S_T), s(S_MASKT)), kind="salu")
    emit_mask(st, X, uid)
    for gap in range(16):
        if gap < 8:
            emit_pv_mfma(st, Y, gap)
            emit_k_read(st, gap)
        else:
            emit_qk_mfma(st, Y, gap - 8)
            emit_v_reads(st, gap - 8)
        if first_dma and 1 <= gap <= 4:
            p = gap - 1
            emit_dma(st, S_KRS if p < 2 else S_VRS, S_KSOFF if p < 2 else S_VSOFF, (S_D0, S_D1, S_D2, S_D3)[p], p & 1, loop=True)
        emit_exp(st, X, 2 * gap)
        emit_exp(st, X, 2 * gap + 1)
        if gap >= 1:
            emit_sum_pack(st, X, gap - 1)
    emit_sum_pack(st, X, 15)
    # tail: the lane's partial row sum must stay inside (2^-64, 2^64); anything else takes the whole wave through the re-centring
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(ACC1)), [ACC0, ACC1], [ACC0])
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(TL), v(x["L"]), v(ACC0)), [x["L"], ACC0], [TL])
    st.valu("v_add_u32_e32 %s, 0xe0800000, %s" % (v(TC), v(TL)), [TL], [TC])
    st.valu("v_cmp_le_u32_e32 vcc, 0x40000000, %s" % v(TC), [TC], [])
    if not ABL:
        st.raw("s_cbranch_vccnz .Lrare_%s_%s" % (X, "%="), kind="salu")
    st.label(".Lret_%s_%s" % (X, "%="))
    st.valu("v_mov_b32_e32 %s, %s" % (v(x["L"]), v(TL)), [TL], [x["L"]])


def emit_rare(st, X):
    """re-centre block X (a change of reference, exact up to rounding) and redo its tile's numerators; returns to .Lret_X with TL = new l."""
    x = blk(X)
    R = KF            # scratch: the K fragment buffer is idle at a phase's tail
    st.label(".Lrare_%s_%s" % (X, "%="))
    S_ = x["S"]
    st.valu("v_max3_f32 %s, %s, %s, %s" % (v(R), v(S_), v(S_ + 1), v(S_ + 2)), [S_, S_ + 1, S_ + 2], [R])
    for i in range(3, 31, 2):
        st.valu("v_max3_f32 %s, %s, %s, %s" % (v(R), v(R), v(S_ + i), v(S_ + i + 1)), [R, S_ + i, S_ + i + 1], [R])
    st.valu("v_max_f32_e32 %s, %s, %s" % (v(R), v(R), v(S_ + 31)), [R, S_ + 31], [R])
    st.drain_lds()
    st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R + 1), v(XADDR), v(R)), kind="lds")
    st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R + 2), v(XADDR), v(x["L"])), kind="lds")
    st.raw("s_waitcnt lgkmcnt(0)", kind="wait")
    st.valu("v_max_f32_e32 %s, %s, %s" % (v(R), v(R), v(R + 1)), [], [])                       # the row's maximum (relative to mref)
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(R + 2), v(R + 2), v(x["L"])), [], [])              # the row's sum so far
    st.valu("v_log_f32_e32 %s, %s" % (v(R + 3), v(R + 2)), [], [])                               # log2; -inf for an empty row
    st.raw("s_nop 1", states=2, kind="nop")
    st.valu("v_max_f32_e32 %s, %s, %s" % (v(R + 4), v(R), v(R + 3)), [], [])                   # delta
    st.valu("v_cmp_lt_f32_e32 vcc, %s, %s" % (v(NINF), v(R + 4)), [], [])                        # false for -inf and NaN
    st.valu("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(R + 4), v(R + 4)), [], [])
    st.valu("v_exp_f32_e64 %s, -%s" % (v(R + 5), v(R + 4)), [], [])                              # alpha
    st.valu("v_cmp_lt_f32_e32 vcc, 0, %s" % v(R + 2), [], [])                                    # row not empty
    st.raw("s_nop 1", states=2, kind="nop")
    st.valu("v_cndmask_b32_e32 %s, 0, %s, vcc" % (v(R + 5), v(R + 5)), [], [])                  # alpha (0 for an empty row: its O and l are 0)
    st.valu("v_mul_f32_e32 %s, %s, %s" % (v(R + 6), v(R + 2), v(R + 5)), [], [])
    st.valu("v_mul_f32_e32 %s, 0.5, %s" % (v(R + 6), v(R + 6)), [], [])                          # the row's rescaled sum, shared evenly by its two lanes
    st.valu("v_add_f32_e32 %s, %s, %s" % (v(x["M"]), v(x["M"]), v(R + 4)), [], [])
    for i in range(16):
        st.valu("v_sub_f32_e32 %s, %s, %s" % (v(x["N"] + i), v(x["N"] + i), v(R + 4)), [], [])
    st.raw("s_nop 7", states=8, kind="nop")
    st.raw("s_nop 7", states=8, kind="nop")
    for i in range(32):
        st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + 8), a(x["O"] + i)))
        st.raw("s_nop 1", states=2, kind="nop")
        st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 8), v(R + 8), v(R + 5)))
        st.raw("s_nop 1", states=2, kind="nop")
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(x["O"] + i), v(R + 8)))
    for i in range(32):
        st.valu("v_sub_f32_e32 %s, %s, %s" % (v(S_ + i), v(S_ + i), v(R + 4)), [S_ + i], [S_ + i])
    for k in range(16):
        st.raw("v_exp_f32_e32 %s, %s" % (v(R + 8), v(S_ + 2 * k)))
        st.raw("v_exp_f32_e32 %s, %s" % (v(R + 9), v(S_ + 2 * k + 1)))
        st.raw("s_nop 1", states=2, kind="nop")
        if k == 0:
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(R + 8), v(R + 9)))
        else:
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(R + 8)))
            st.raw("v_add_f32_e32 %s, %s, %s" % (v(ACC0), v(ACC0), v(R + 9)))
        st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(x["P"] + k), v(R + 8), v(R + 9)))
    st.raw("v_add_f32_e32 %s, %s, %s" % (v(TL), v(R + 6), v(ACC0)))
    st.raw("s_branch .Lret_%s_%s" % (X, "%="))


def emit_step_salu(st):
    """ring slots and request offsets of step t (s58)."""
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_T)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_KCUR), s(S_TMP)), kind="salu")                  # slot of K(t) and V(t)
    st.raw("s_mov_b32 %s, %s" % (s(S_VCUR), s(S_KCUR)), kind="salu")
    st.raw("s_add_u32 %s, %s, 1" % (s(S_TMP), s(S_T)), kind="salu")
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_TMP)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_KNXT), s(S_TMP)), kind="salu")                  # slot of K(t+1)
    # requests of this step: K(t+3) -> slot (t+3)&3, V(t+2) -> slot (t+2)&3 ; tiles past the end: an out-of-range offset (counted, no fetch)
    st.raw("s_add_u32 %s, %s, 3" % (s(S_TMP), s(S_T)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_KSOFF), s(S_TMP)), kind="salu")
    st.raw("s_cmp_lt_u32 %s, %s" % (s(S_TMP), s(S_NT)), kind="salu")
    st.raw("s_cselect_b32 %s, %s, 0x7f000000" % (s(S_KSOFF), s(S_KSOFF)), kind="salu")
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_TMP)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_TMP), s(S_TMP)), kind="salu")
    st.raw("s_add_u32 %s, %s, %s" % (s(S_D0), s(S_KDST), s(S_TMP)), kind="salu")
    st.raw("s_add_u32 %s, %s, 0x400" % (s(S_D1), s(S_D0)), kind="salu")
    st.raw("s_add_u32 %s, %s, 2" % (s(S_TMP), s(S_T)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_VSOFF), s(S_TMP)), kind="salu")
    st.raw("s_cmp_lt_u32 %s, %s" % (s(S_TMP), s(S_NT)), kind="salu")
    st.raw("s_cselect_b32 %s, %s, 0x7f000000" % (s(S_VSOFF), s(S_VSOFF)), kind="salu")
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_TMP)), kind="salu")
    st.raw("s_lshl_b32 %s, %s, 13" % (s(S_TMP), s(S_TMP)), kind="salu")
    st.raw("s_add_u32 %s, %s, %s" % (s(S_D2), s(S_VDST), s(S_TMP)), kind="salu")
    st.raw("s_add_u32 %s, %s, 0x400" % (s(S_D3), s(S_D2)), kind="salu")
    # keys of tile t that exist: k_len - 64 t (the mask code runs only where this is < 64)
    st.raw("s_lshl_b32 %s, %s, 6" % (s(S_TMP), s(S_T)), kind="salu")
    st.raw("s_sub_u32 %s, %s, %s" % (s(S_REM), s(S_KL), s(S_TMP)), kind="salu")


def build(drop):
    st = Stream()
    # ---- prologue ----------------------------------------------------------------------------------------------------------------
    st.comment("==== attention forward v4 (%s): generated by tools/gen_attn_fwd4.py - do not edit" % ("train: dropout" if drop else "eval"))
    st.comment("---- inputs into the block's own registers")
    for name, reg in (("voff0", VOFF), ("voff1", VOFF + 1), ("kofs0", KB), ("kofs1", KB + 1), ("kofs2", KB + 2), ("kofs3", KB + 3),
                      ("vofs0", VB), ("vofs1", VB + 1), ("vofs2", VB + 2), ("vofs3", VB + 3), ("qoff", QOFF), ("ooff", OOFF),
                      ("lseoff", LSEOFF), ("thr", THR)):
        st.raw("v_mov_b32_e32 %s, %%[%s]" % (v(reg), name))
    st.raw("v_xor_b32_e32 %s, 32, %%[lane]" % v(XADDR))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(XADDR), v(XADDR)))
    st.raw("v_lshlrev_b32_e32 %s, 4, %%[lane]" % v(TC))                  # lane * 16: this lane's bytes of a 1-KiB zero-fill row
    st.raw("v_mov_b32_e32 %s, 0xff800000" % v(NINF))
    for name, reg in (("kb", S_KRS), ("vb", S_VRS), ("qb", S_QRS), ("cb", S_CRS), ("lb", S_LRS)):
        st.raw("s_mov_b64 %s, %%[%s]" % (s(reg, 2), name))
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(reg + 1), s(reg + 1)))
        st.raw("s_mov_b32 %s, 0x00020000" % s(reg + 3))
    for name, reg in (("kl", S_KL), ("nt", S_NT), ("kdst", S_KDST), ("dsc", S_DSC), ("hrow", S_HROW), ("lq", S_LQ), ("csize", S_CSIZE)):
        st.raw("s_mov_b32 %s, %%[%s]" % (s(reg), name))
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_KRS + 2), s(S_KL)))             # K and V: k_len rows of 128 bytes (rows past it read as zeros)
    st.raw("s_mov_b32 %s, %s" % (s(S_VRS + 2), s(S_KRS + 2)))
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_QRS + 2), s(S_LQ)))
    st.raw("s_mov_b32 %s, %s" % (s(S_CRS + 2), s(S_CSIZE)))
    st.raw("s_mov_b32 %s, %%[lsz]" % s(S_LRS + 2))
    st.raw("s_add_u32 %s, %s, 0x8000" % (s(S_VDST), s(S_KDST)))
    st.raw("s_and_b32 %s, %s, 63" % (s(S_TMP), s(S_KL)))                  # the step that holds a ragged last tile (none: -1)
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_MASKT), s(S_NT)))
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_TMP))
    st.raw("s_cselect_b32 %s, -1, %s" % (s(S_MASKT), s(S_MASKT)))
    st.comment("---- Q fragments of both blocks (block B: 32 rows = 4096 bytes further), then K0 V0 K1 V1 K2 by LDS-DMA")
    st.raw("v_add_u32_e32 %s, 0x1000, %s" % (v(TL), v(QOFF)))
    st.raw("s_nop 4")
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(KF + 4 * ks, 4), v(QOFF), s(S_QRS, 4), 32 * ks))
    for ks in range(4):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(KF + 16 + 4 * ks, 4), v(TL), s(S_QRS, 4), 32 * ks))
    for tile, which in ((0, "K"), (0, "V"), (1, "K"), (1, "V"), (2, "K")):
        base = S_KDST if which == "K" else S_VDST
        for piece in range(2):
            st.raw("s_add_u32 %s, %s, 0x%x" % (s(S_TMP), s(base), tile * 8192 + piece * 1024))
            st.raw("s_mov_b32 %s, 0x%x" % (s(S_TMP2), tile * 8192))
            st.raw("s_cmp_lt_u32 %d, %s" % (tile, s(S_NT)))
            st.raw("s_cselect_b32 %s, %s, 0x7f000000" % (s(S_TMP2), s(S_TMP2)))
            emit_dma(st, S_KRS if which == "K" else S_VRS, S_TMP2, S_TMP, piece)
    st.comment("---- state: O = 0, -mref = 0, l = 0, P of block B = 0, V ring slot 3 = 0 (the first phase multiplies it by that P)")
    for i in range(64):
        st.raw("v_accvgpr_write_b32 %s, 0" % a(i))
    for r in rng(NA, 32) + rng(PB, 16) + [LA, LB, MA, MB]:
        st.raw("v_mov_b32_e32 %s, 0" % v(r))
    for r in range(4):
        st.raw("v_mov_b32_e32 %s, 0" % v(T + r))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(TC), s(S_VDST), v(TC)))
    st.raw("ds_write_b128 %s, %s offset:%d" % (v(TC), v(T, 4), 3 * 8192))
    st.raw("ds_write_b128 %s, %s offset:%d" % (v(TC), v(T, 4), 3 * 8192 + 1024))
    st.comment("---- Q into the accumulator file (B operands of every K.Q^T)")
    st.raw("s_waitcnt vmcnt(10)")
    for i in range(32):
        st.raw("v_accvgpr_write_b32 %s, %s" % (a(QA + i), v(KF + i)))
    st.raw("s_waitcnt vmcnt(8) lgkmcnt(0)")        # K0 landed (V0 K1 V1 K2 may be in flight); the zero fill is in LDS
    st.raw("s_barrier")
    st.comment("---- K.Q^T of block A's tile 0; V fragments of the (zero) tile -1")
    st.raw("s_mov_b32 %s, 0" % s(S_T))
    emit_step_salu(st)
    for i in range(4):
        st.valu("v_mov_b32_e32 %s, %s" % (v(KAD + i), v(KB + i)), [], [])
        st.valu("v_add_u32_e32 %s, 0x%x, %s" % (v(VAD + i), 3 * 8192, v(VB + i)), [], [])
    st.raw("s_nop 7", states=8)
    for f in range(8):
        emit_k_read(st, f)
    for f in range(8):
        emit_qk_mfma(st, "A", f)
    for f in range(8):
        emit_v_reads(st, f)
    st.raw("s_cmp_eq_u32 %s, 0" % s(S_NT))
    st.raw("s_cbranch_scc1 .Lfinal_%=")
    # ---- the loop: two passes, the first only to learn the hazard / queue state at the back edge ---------------------------------------
    entry_lds = [set(x) for x in st.lds]
    body_start = len(st.out)
    n_start, mf_start = st.n, dict(st.mfma_w)
    for final_pass in (False, True):
        if final_pass:
            # re-seed with the end-of-body state (relative distances preserved)
            shift = st.n - n_start
            seeded = {r: w - shift for r, w in st.mfma_w.items()}
            end_lds = [set(x) for x in st.lds]
            assert [sorted(x) for x in end_lds] == [sorted(x) for x in entry_lds], "LDS queue at the back edge differs from the entry's"
            del st.out[body_start:]
            st.n = n_start
            st.mfma_w = {r: max(w, seeded.get(r, -10 ** 9)) for r, w in mf_start.items()}
            for r, w in seeded.items():
                st.mfma_w.setdefault(r, w)
            st.lds = [set(x) for x in entry_lds]
            st.nops = 0
            st.counts = {}
        st.label(".Lstep_%=")
        if not (ABL & 16):
            st.raw("s_waitcnt vmcnt(4)", kind="wait")      # K(t+1) and V(t) have landed: everything but the previous step's four requests
            st.raw("s_barrier", kind="salu")
        emit_phase(st, "A", "B", "a", first_dma=True)
        emit_phase(st, "B", "A", "b")
        st.raw("s_add_u32 %s, %s, 1" % (s(S_T), s(S_T)), kind="salu")
        emit_step_salu(st)
        st.raw("s_cmp_lt_u32 %s, %s" % (s(S_T), s(S_NT)), kind="salu")
        st.raw("s_cbranch_scc1 .Lstep_%=", kind="salu")
    body_counts = dict(st.counts)
    body_nops = st.nops
    # ---- after the last step: block B's last P.V ---------------------------------------------------------------------------------------
    for f in range(8):
        emit_pv_mfma(st, "B", f)
    st.label(".Lfinal_%=")
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")             # (out-of-range requests of the last steps: nothing may land after the LDS is released)
    st.raw("s_nop 7", states=8)
    st.raw("s_nop 7", states=8)
    st.comment("---- epilogue: O / l -> bf16 rows of ctx; lse = mref + log2(l)")
    R = KF
    for X in ("A", "B"):
        x = blk(X)
        st.raw("ds_bpermute_b32 %s, %s, %s" % (v(R), v(XADDR), v(x["L"])))
        st.raw("s_waitcnt lgkmcnt(0)")
        st.raw("v_add_f32_e32 %s, %s, %s" % (v(R), v(R), v(x["L"])))
        st.raw("v_rcp_f32_e32 %s, %s" % (v(R + 1), v(R)))
        st.raw("v_log_f32_e32 %s, %s" % (v(R + 2), v(R)))
        st.raw("s_nop 1", states=2)
        if drop:
            st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 1), s(S_DSC), v(R + 1)))
        st.raw("v_add_f32_e32 %s, %s, %s" % (v(R + 2), v(R + 2), v(x["M"])))
        if X == "B":
            st.raw("v_add_u32_e32 %s, %s, %s" % (v(OOFF), s(S_HROW), v(OOFF)))
            st.raw("v_add_u32_e32 %s, 0x80, %s" % (v(LSEOFF), v(LSEOFF)))
        for dt in range(2):
            for g in range(4):
                for j in range(4):
                    st.raw("v_accvgpr_read_b32 %s, %s" % (v(R + 4 + j), a(x["O"] + 16 * dt + 4 * g + j)))
                st.raw("s_nop 0")
                for j in range(4):
                    st.raw("v_mul_f32_e32 %s, %s, %s" % (v(R + 4 + j), v(R + 4 + j), v(R + 1)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 8 + 2 * (4 * dt + g)), v(R + 4), v(R + 5)))
                st.raw("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(R + 9 + 2 * (4 * dt + g)), v(R + 6), v(R + 7)))
        st.raw("s_nop 1", states=2)
        for dt in range(2):
            for g in range(4):
                st.raw("buffer_store_dwordx2 %s, %s, %s, 0 offen offset:%d" % (v(R + 8 + 2 * (4 * dt + g), 2), v(OOFF), s(S_CRS, 4), 64 * dt + 16 * g))
        st.raw("s_mov_b32 exec_hi, 0")                  # lanes 0..31 hold one row each
        st.raw("buffer_store_dword %s, %s, %s, 0 offen" % (v(R + 2), v(LSEOFF), s(S_LRS, 4)))
        st.raw("s_mov_b32 exec_hi, -1")
    st.raw("s_waitcnt vmcnt(0)")
    st.raw("s_endpgm")
    emit_rare(st, "A")
    emit_rare(st, "B")
    return st, body_counts, body_nops


def main():
    global ABL
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "end-to-end_asr_pytorch_amd", "csrc", "attention_fwd4_asm.inc")
    if len(sys.argv) > 2 and sys.argv[1] == "--abl":        # tools/ab_attn4.sh: python tools/gen_attn_fwd4.py --abl 3 /tmp/x.inc
        ABL = int(sys.argv[2])
        out = sys.argv[3]
    with open(out, "w") as f:
        f.write("// generated by tools/gen_attn_fwd4.py - do not edit (edit the generator and run it again)\n")
        for drop in (False,):
            st, counts, nops = build(drop)
            f.write("#define ATTN4_ASM_%s \\\n" % ("TRAIN" if drop else "EVAL"))
            for line in st.out:
                f.write('    "%s\\n" \\\n' % line.replace("\\", "\\\\").replace('"', '\\"'))
            f.write('    ""\n')
            sys.stderr.write("%s: %d lines; per step: %s; s_nop states padded in the loop: %d\n" % ("train" if drop else "eval", len(st.out), counts, nops))
        regs = ["v%d" % i for i in range(232)] + ["a%d" % i for i in range(96)] + ["s%d" % i for i in range(34, 100)] + ["vcc", "memory"]
        f.write("#define ATTN4_ASM_CLOBBERS %s\n" % ", ".join('"%s"' % r for r in regs))


if __name__ == "__main__":
    main()
