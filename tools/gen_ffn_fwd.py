#!/usr/bin/env python3
"""Generator of the encoder feed-forward sub-layer's forward loop (csrc/ffn_fwd2_asm.inc, included by ffn2.hip).

    y = LayerNorm(dropout(relu(x W1^T + b1) W2^T + b2) + x)       src/transformer/module.py:48-53  (d_model 256, d_ff = 64 NC)

Why generated assembly (profiles/r5/ffn_ablation.txt): at ONE wave per SIMD the loop's three parts simply add up - 27.5 us of MFMAs (the
matrix pipe's whole time), 16 us of LDS-DMA issue, 33 us of prologue / ReLU / barriers / epilogue - because a lone wave issues one
instruction of any kind at a time.  A second wave per SIMD overlaps them, but only inside 256 registers per wave, which hipcc's own
schedule of the C++ form does not reach (84.6 us against 73.0).  This script fixes every register and places every instruction.

Structure: a workgroup = 128 tokens = 8 waves; waves wv and wv + 4 sit on one SIMD and share the 32 tokens of pair p = wv & 3, half
w = wv >> 2.  Per 64-unit chunk i of the hidden dimension and wave:
  first product   S^T[32 units of half w x 32 tok] = W1c . X^T    16 MFMAs (K = 256, X in 64 VGPRs as B operands, accumulator starts at b1)
  ReLU + bf16     in place (v_cvt_pk_bf16_f32 + v_pk_max_i16) = two B fragments of the second product; written to the pair's LDS tile
                  [32 tok][64 units] for the partner (and for the whole-line global stores of the hidden activation, training)
  second product  Y^T[128 rows of half w x 32 tok] += W2c . H^T   16 MFMAs (K = 64: own two k-steps from registers, the partner's two
                  from the tile), Y in 64 AGPRs
An iteration runs first(i) then second(i - 1): ReLU(i), the tile write, the mask word and the stores sit in the shadow of second(i - 1)'s
MFMAs; ONE workgroup barrier per chunk covers both the weight ring (LDS-DMA double buffers, as in ffn.hip) and the H exchange.
The loop is unrolled twice (buffer parities as immediates).  Everything lane-dependent arrives as parameters the C++ prologue left in
LDS (one dword per thread and parameter), so the block needs no VGPR inputs.

Registers: v0-v175, a0-a63, s34-s99 (m0 saved / restored).  Inputs: the kernel-argument pointer, the wave number, the LDS array's address.
"""
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from asmgen import Stream, v, a, s, write_inc      # noqa: E402

TRAIN = True
DMA_POLICY = ""     # cache-policy modifiers of the LDS-DMA loads ("nt", "sc0", "sc1", combinations): --policy
ABL = 0         # timing-only builds (results garbage): 1 no MFMA, 2 no ReLU / mask VALU, 4 no fragment reads, 8 no LDS-DMA, 16 no hid / mask stores,
                # 32 no bias, 128 only the W1 half of the LDS-DMA; 64 (training form): s_memtime / s_memrealtime around the loop, left in the
                # first 16 bytes of every 16-byte piece of the hidden activation's first chunk (t0, r0, t1, r1; tools/check_ffn2.py --stamps)

# ---- VGPRs -----------------------------------------------------------------------------------------------------------------------
X = 0               # 16 B fragments (4 registers each): X[token r][16 ks + 8 h .. + 8]
S = 64              # first product's accumulator
BI = 80             # the next chunk's bias in accumulator layout (SrcC of the chunk's first MFMA)
HO = (96, 104)      # own H fragments (2 k-steps x 4) by chunk parity
HP = 112            # the partner's two fragments
HOUT = 120          # two 16-byte row pieces of the previous chunk's tile on their way to global memory
FR = 128            # fragment ring: 4 slots
A1 = 144            # 8 LDS addresses of the W1 fragment reads (k-step & 7)
A2O, A2P = 152, 154     # W2 fragment read addresses of the own / partner's k-steps (g = 0, 1)
OFF1, OFF2 = 156, 160   # LDS-DMA source offsets of this wave's 4 + 4 pieces
HXW, HXP, HXR = 164, 166, 168
HOFF = 170
BOFF, WORD, TMP, TMP2 = 172, 173, 174, 175
NV = 176
Y = 0               # AGPR: 4 row tiles x 16
NA = 64
# ---- SGPRs (the block owns s34-s99) ------------------------------------------------------------------------------------------------
S_M0SAVE, S_TMP = 34, 35
S_W1RS, S_W2RS, S_HRS, S_BRS, S_B1 = 36, 40, 44, 48, 52           # (resource quads 4-aligned, pairs even)
S_NC, S_I, S_W1SOFF, S_W2SOFF, S_HSOFF, S_BSOFF, S_BSTRIDE, S_W1DST, S_W1LAST, S_YBASE, S_WV, S_K8000, S_KSEL, S_BI = range(54, 68)
SB = 68             # 32 SGPRs: the next chunk's 32 bias values of this wave's half (units 32 w ..); S_B1 points at chunk S_BI's
S_LO, S_HI = 34, 99
# prologue only, inside the bias block (dead before the first bias request)
S_M, S_DFF, S_MP, S_PBASE, S_SM0, S_KA, S_XRS = 68, 70, 71, 92, 93, 94, 96
# kernel-argument block (ffn2.hip: Ffn2Args; static_asserts there)
KA_X16, KA_W1, KA_B1, KA_W2, KA_HID, KA_BITS, KA_M = 0, 16, 24, 32, 72, 80, 128

PARAMS = ["a1_0", "a1_1", "a1_2", "a1_3", "a1_4", "a1_5", "a1_6", "a1_7", "a2o_0", "a2o_1", "a2p_0", "a2p_1", "off1_0", "off1_1", "off1_2",
          "off1_3", "off2_0", "off2_1", "off2_2", "off2_3", "hxw_0", "hxw_1", "hxp_0", "hxp_1", "hxr_0", "hxr_1", "hoff_0", "hoff_1", "boff",
          "xoff"]
PARAM_REG = dict(zip(PARAMS, list(range(A1, A1 + 8)) + [A2O, A2O + 1, A2P, A2P + 1] + list(range(OFF1, OFF1 + 4)) + list(range(OFF2, OFF2 + 4)) +
                     [HXW, HXW + 1, HXP, HXP + 1, HXR, HXR + 1, HOFF, HOFF + 1, BOFF, TMP]))

# where an iteration's eight LDS-DMA requests sit, by role (role = wave >> 2: the two waves of a SIMD).  An LDS-DMA instruction holds
# its wave's issue for 60-180 cycles; with both waves of a SIMD at the same gap (r6 first form: one code path) neither has an MFMA to
# issue meanwhile - the loop ran 52 us for 27 us of MFMAs and lost 13 us without the requests.  The roles' gaps interleave.
DMA_GAPS = {0: (0, 3, 6, 9, 12, 15, 18, 21), 1: (1, 4, 7, 10, 13, 16, 19, 22)}
DMA_GAPS_FIRST = {0: (0, 2, 4, 6, 8, 10, 12, 14), 1: (1, 3, 5, 7, 9, 11, 13, 15)}
STORE_GAPS = {0: (22, 24), 1: (23, 25)}       # behind the iteration's last request: the closing wait then leaves the stores in flight
BIAS_GAP0 = 2           # the 16 bias registers: two per gap from here
RELU_GAP0 = 22


def V(base, n=1):
    return ["v%d" % i for i in range(base, base + n)]


def A(base, n=1):
    return ["a%d" % i for i in range(base, base + n)]


# ---- pieces ----------------------------------------------------------------------------------------------------------------------
def frag_read(st, f, q, first_only):
    """fragment f (0..31) of an iteration with chunk parity q into ring slot f & 3: f < 16 = W1 k-step f (buffer q), else W2 fragment
    (buffer q ^ 1): k-step group (f - 16) >> 2 (own 0, own 1, partner 0, partner 1), row tile (f - 16) & 3"""
    if ABL & 4:
        return
    dst = FR + 4 * (f & 3)
    if f < 16:
        st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(dst, 4), v(A1 + (f & 7)), q * 32768 + (f >> 3) * 256), V(A1 + (f & 7)), V(dst, 4))
    else:
        assert not first_only
        kk = f - 16
        g, ytl = kk >> 2, kk & 3
        ad = (A2O + g) if g < 2 else (A2P + g - 2)
        st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(dst, 4), v(ad), (q ^ 1) * 32768 + ytl * 4096), V(ad), V(dst, 4))


def mfma_first(st, k):
    fr = FR + 4 * (k & 3)
    if ABL & 1:
        st._need(set(V(fr, 4)))
        return
    c = v(BI, 16) if k == 0 else v(S, 16)
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (v(S, 16), v(fr, 4), v(X + 4 * k, 4), c),
            V(fr, 4) + V(X + 4 * k, 4), V(BI, 16) if k == 0 else [], V(S, 16))


def mfma_second(st, kk, q):
    """second product of the PREVIOUS chunk (parity q ^ 1): k-step group g = kk >> 2 (own 0, own 1, partner 0, partner 1), row tile kk & 3"""
    fr = FR + 4 * (kk & 3)
    g, ytl = kk >> 2, kk & 3
    b = (HO[q ^ 1] + 4 * g) if g < 2 else (HP + 4 * (g - 2))
    if ABL & 1:
        st._need(set(V(fr, 4)) | set(V(b, 4)))
        return
    st.mfma("v_mfma_f32_32x32x16_bf16 %s, %s, %s, %s" % (a(Y + 16 * ytl, 16), v(fr, 4), v(b, 4), a(Y + 16 * ytl, 16)),
            V(fr, 4) + V(b, 4), [], A(Y + 16 * ytl, 16))


def dma_piece(j, q):
    """LDS-DMA request j (0..7) of an iteration with parity q: 0..3 = W1 of the NEXT chunk into buffer q ^ 1, 4..7 = W2 of THIS chunk
    into buffer q.  -> (the M0 instruction, (the load, its address register)): one other instruction has to sit between the two."""
    if j < 4:
        m0 = "s_add_u32 m0, %s, 0x%x" % (s(S_W1DST), (q ^ 1) * 32768 + j * 1024)
        ld = ("buffer_load_dwordx4 %s, %s, %s offen%s lds" % (v(OFF1 + j), s(S_W1RS, 4), s(S_W1SOFF), (" " + DMA_POLICY) if DMA_POLICY else ""), V(OFF1 + j))
    else:
        m0 = "s_add_u32 m0, %s, 0x%x" % (s(S_W1DST), 0x10000 + q * 32768 + (j - 4) * 1024)
        ld = ("buffer_load_dwordx4 %s, %s, %s offen%s lds" % (v(OFF2 + j - 4), s(S_W2RS, 4), s(S_W2SOFF), (" " + DMA_POLICY) if DMA_POLICY else ""), V(OFF2 + j - 4))
    if ABL & 256:       # (timing probe: the same bytes into registers instead of LDS)
        off, rs, so = (OFF1 + j, S_W1RS, S_W1SOFF) if j < 4 else (OFF2 + j - 4, S_W2RS, S_W2SOFF)
        ld = ("buffer_load_dwordx4 %s, %s, %s, %s offen %s" % (v(HOUT + 4 * (j & 1), 4), v(off), s(rs, 4), s(so), DMA_POLICY), V(off))
    return m0, ld


def bias_lo(st, j):
    """BI[j] <- the bias of the unit accumulator register j holds in lane half 0: unit {0,4,16,20}[j >> 2] + (j & 3) of the wave's 32"""
    st.valu("v_mov_b32_e32 %s, %s" % (v(BI + j), s(SB + (0, 4, 16, 20)[j >> 2] + (j & 3))), [], V(BI + j))


def bias_hi(st):
    """... and in lane half 1: eight units further.  (A v_cndmask with a scalar source and vcc breaks the one-scalar-operand rule; the
    upper half is written under exec = lanes 32-63 instead: one block, nothing else may sit inside it.)"""
    st.salu("s_mov_b32 exec_lo, 0")
    for j in range(16):
        st.valu("v_mov_b32_e32 %s, %s" % (v(BI + j), s(SB + 8 + (0, 4, 16, 20)[j >> 2] + (j & 3))), [], V(BI + j))
    st.salu("s_mov_b32 exec_lo, -1")


def bias_advance(st):
    """S_B1 -> the next chunk's 32 values, staying on the last chunk's"""
    st.salu("s_add_u32 %s, %s, 1" % (s(S_TMP), s(S_BI)))
    st.salu("s_cmp_lt_u32 %s, %s" % (s(S_TMP), s(S_NC)))
    st.salu("s_cselect_b32 %s, 1, 0" % s(S_TMP))
    st.salu("s_add_u32 %s, %s, %s" % (s(S_BI), s(S_BI), s(S_TMP)))
    st.salu("s_lshl_b32 %s, %s, 8" % (s(S_TMP), s(S_TMP)))
    st.salu("s_add_u32 %s, %s, %s" % (s(S_B1), s(S_B1), s(S_TMP)))
    st.salu("s_addc_u32 %s, %s, 0" % (s(S_B1 + 1), s(S_B1 + 1)))


def bias_request(st):
    """the 32 values of the chunk S_B1 points at; out of order with the LDS queue, which only makes the counted LDS waits conservative -
    consumed behind the iteration's closing lgkmcnt(0)"""
    st.salu("s_load_dwordx16 %s, %s, 0x0" % (s(SB, 16), s(S_B1, 2)))
    st.salu("s_load_dwordx16 %s, %s, 0x40" % (s(SB + 16, 16), s(S_B1, 2)))


def relu_pair(st, p, q):
    """register pair p (0..7) of S -> two bf16 in HO[q][p], ReLU as a signed 16-bit max; training: the pair's two mask bits"""
    if ABL & 2:
        return
    ho = HO[q] + p
    st.valu("v_cvt_pk_bf16_f32 %s, %s, %s" % (v(ho), v(S + 2 * p), v(S + 2 * p + 1)), V(S + 2 * p, 2), V(ho))
    st.valu("v_pk_max_i16 %s, %s, 0" % (v(ho), v(ho)), V(ho), V(ho))
    if TRAIN:
        # bit 15 / 31 of w + 0x7fff7fff: that half of w is not zero (ReLU-ed halves have their sign bits clear); the word shifts right once
        # per pair, so pair p ends at bits 8 + p (even unit) and 24 + p (odd unit)
        st.valu("v_add_u32_e32 %s, 0x7fff7fff, %s" % (v(TMP), v(ho)), V(ho), V(TMP))
        if p:
            st.valu("v_lshrrev_b32_e32 %s, 1, %s" % (v(WORD), v(WORD)), V(WORD), V(WORD))
            st.valu("v_and_or_b32 %s, %s, %s, %s" % (v(WORD), v(TMP), s(S_K8000), v(WORD)), V(TMP) + V(WORD), V(WORD))
        else:
            st.valu("v_and_b32_e32 %s, %s, %s" % (v(WORD), s(S_K8000), v(TMP)), V(TMP), V(WORD))


def h_write(st, g, q):
    st.lds_write("ds_write_b128 %s, %s offset:%d" % (v(HXW + g), v(HO[q] + 4 * g, 4), q * 16384), V(HXW + g) + V(HO[q] + 4 * g, 4))


def emit_iteration(st, q, first, last, role):
    """first: chunk 0 (no second product, nothing of a previous chunk to read or store); last: the pseudo-iteration behind the last chunk
    (second product of chunk NC - 1 only)."""
    assert not (first and last)
    # ---- behind the barrier: the partner's H fragments and this wave's rows of the previous chunk's tile; the ring's first four -------
    if not first:
        for g in range(2):
            st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(HP + 4 * g, 4), v(HXP + g), (q ^ 1) * 16384), V(HXP + g), V(HP + 4 * g, 4))
        if TRAIN and not (ABL & 16):
            for ps in range(2):
                st.lds_read("ds_read_b128 %s, %s offset:%d" % (v(HOUT + 4 * ps, 4), v(HXR + ps), (q ^ 1) * 16384), V(HXR + ps), V(HOUT + 4 * ps, 4))
    f0 = 16 if last else 0
    nf = 16 if first else 32
    for f in range(f0, f0 + 4):
        frag_read(st, f, q, first)
    dma = [dma_piece(j, q) for j in range(8)] if not last and not (ABL & 8) else []
    dma_gaps = (DMA_GAPS_FIRST if first else DMA_GAPS)[role]
    store_gaps = STORE_GAPS[role] if not last else (20, 21)
    for k in range(f0, nf):
        # ---- the MFMA of this step ----
        if k < 16:
            mfma_first(st, k)
        else:
            mfma_second(st, k - 16, q)
        nxt = k + 4
        has_read = nxt < nf
        # ---- fillers ----
        piece = dma_gaps.index(k) if (dma and k in dma_gaps) else -1
        if (ABL & 128) and piece >= 4:
            piece = -1
        if piece >= 0:
            st.salu(dma[piece][0])
        if has_read:
            frag_read(st, nxt, q, first)
        elif piece >= 0:
            st.raw("s_nop 0", kind="nop")
        if piece >= 0:
            st.vmem(dma[piece][1][0], dma[piece][1][1], [], tag="dma", kind="dma")
        if not last and not (ABL & 32) and BIAS_GAP0 <= k < BIAS_GAP0 + 8:           # the next chunk's bias (BI was SrcC of MFMA 0): out of the scalar registers
            bias_lo(st, 2 * (k - BIAS_GAP0))
            bias_lo(st, 2 * (k - BIAS_GAP0) + 1)
        if not last and not (ABL & 32) and k == BIAS_GAP0 + 8:
            bias_hi(st)
            bias_request(st)
        # the previous chunk's rows leave (training): two whole-line stores
        if TRAIN and not first and not (ABL & 16) and k in store_gaps:
            ps = store_gaps.index(k)
            st.vmem("buffer_store_dwordx4 %s, %s, %s, %s offen" % (v(HOUT + 4 * ps, 4), v(HOFF + ps), s(S_HRS, 4), s(S_HSOFF)),
                    V(HOUT + 4 * ps, 4) + V(HOFF + ps), [], tag="store", kind="store")
        # ReLU / pack / mask of THIS chunk in the second product's shadow
        if not last and not first and RELU_GAP0 <= k < RELU_GAP0 + 8:
            relu_pair(st, k - RELU_GAP0, q)
            if k - RELU_GAP0 == 3:
                h_write(st, 0, q)
            if k - RELU_GAP0 == 7:
                h_write(st, 1, q)
    if first:       # chunk 0: right behind the first product
        for p in range(8):
            relu_pair(st, p, q)
            if p == 3:
                h_write(st, 0, q)
        h_write(st, 1, q)
    if not last and TRAIN and not (ABL & 18):
        # the chunk's 16 mask bits of this lane: bytes 1 and 3 of the word -> one short; [chunk][half w][lane half h][token]
        st.valu("v_perm_b32 %s, %s, %s, %s" % (v(WORD), v(WORD), v(WORD), s(S_KSEL)), V(WORD), V(WORD))
        st.vmem("buffer_store_short %s, %s, %s, %s offen" % (v(WORD), v(BOFF), s(S_BRS, 4), s(S_BSOFF)), V(WORD) + V(BOFF), [], tag="store", kind="store")
    if last:
        return
    # ---- end of the iteration: the chunk counter and the offsets that follow it; this iteration's LDS-DMA has landed, the tile writes are
    # done, the bias request is back; stores may stay in flight ----
    st.salu("s_add_u32 %s, %s, 1" % (s(S_I), s(S_I)))
    st.salu("s_add_u32 %s, %s, 0x8000" % (s(S_W1SOFF), s(S_W1SOFF)))
    st.salu("s_min_u32 %s, %s, %s" % (s(S_W1SOFF), s(S_W1SOFF), s(S_W1LAST)))
    st.salu("s_add_u32 %s, %s, 128" % (s(S_W2SOFF), s(S_W2SOFF)))
    bias_advance(st)
    if not first:
        st.salu("s_add_u32 %s, %s, 128" % (s(S_HSOFF), s(S_HSOFF)))
    st.salu("s_add_u32 %s, %s, %s" % (s(S_BSOFF), s(S_BSOFF), s(S_BSTRIDE)))
    st.vm_wait({"dma"}, with_lds=True)
    st.salu("s_barrier")
    st.salu("s_cmp_eq_u32 %s, %s" % (s(S_I), s(S_NC)))


def emit_role(st, role):
    """chunk 0, the loop (two chunks per trip), the two tails: for one of the two waves of a SIMD"""
    U = "%d_%%=" % role
    emit_iteration(st, 0, True, False, role)
    st.salu("s_cbranch_scc1 .Ltail1_" + U)
    # the loop: generated twice, the first time only to learn the state at the back edge
    snap = st.snapshot()
    for final in (False, True):
        if final:
            assert [sorted(x) for x in st.lds] == [sorted(x) for x in snap["lds"]], "LDS queue at the back edge differs from the entry's"
            vm_back = list(st.vm)
            st.rewind(snap)
            st.vm = vm_back          # stores of the previous trip still in flight (vmcnt is counted against what the body itself issues)
        st.label(".Ltrip_" + U)
        emit_iteration(st, 1, False, False, role)
        st.salu("s_cbranch_scc1 .Ltail0_" + U)
        emit_iteration(st, 0, False, False, role)
        st.salu("s_cbranch_scc0 .Ltrip_" + U)
    counts, nops = dict(st.counts), st.nops
    loop_exit = st.snapshot()
    for q in (1, 0):
        st.label(".Ltail%d_%s" % (q, U))
        emit_iteration(st, q, False, True, role)
        st.salu("s_branch .Lend_%=")
        st.lds, st.vm, st.mfma_w, st.n = [set(x) for x in loop_exit["lds"]], list(loop_exit["vm"]), dict(loop_exit["mfma"]), loop_exit["n"]
    return counts, nops


def build():
    st = Stream()
    U = "%="
    st.comment("==== feed-forward forward loop (%s): generated by tools/gen_ffn_fwd.py" % ("training" if TRAIN else "eval"))
    st.raw("s_mov_b32 %s, m0" % s(S_M0SAVE))
    st.raw("s_mov_b64 %s, %%[ka]" % s(S_KA, 2))
    st.raw("s_mov_b32 %s, %%[wv]" % s(S_WV))
    st.raw("s_mov_b32 %s, %%[smem0]" % s(S_SM0))
    # ---- the kernel's arguments ----
    for off, reg in ((KA_X16, S_XRS), (KA_W1, S_W1RS), (KA_W2, S_W2RS), (KA_B1, S_B1), (KA_HID, S_HRS), (KA_BITS, S_BRS)):
        st.raw("s_load_dwordx2 %s, %s, 0x%x" % (s(reg, 2), s(S_KA, 2), off))
    st.raw("s_load_dwordx4 %s, %s, 0x%x" % (s(S_M, 4), s(S_KA, 2), KA_M))          # M, L, dff, Mp
    st.raw("s_waitcnt lgkmcnt(0)")
    for reg in (S_XRS, S_W1RS, S_W2RS, S_HRS, S_BRS):
        st.raw("s_and_b32 %s, %s, 0xffff" % (s(reg + 1), s(reg + 1)))
        st.raw("s_mov_b32 %s, 0x00020000" % s(reg + 3))
    st.raw("s_lshr_b32 %s, %s, 6" % (s(S_NC), s(S_DFF)))
    st.raw("s_lshl_b32 %s, %s, 9" % (s(S_XRS + 2), s(S_M)))                 # x16: M rows of 512 bytes
    st.raw("s_lshl_b32 %s, %s, 9" % (s(S_W1RS + 2), s(S_DFF)))              # W1, W2: dff * 256 bf16
    st.raw("s_mov_b32 %s, %s" % (s(S_W2RS + 2), s(S_W1RS + 2)))
    if TRAIN:
        st.raw("s_mul_i32 %s, %s, %s" % (s(S_HRS + 2), s(S_M), s(S_DFF)))
        st.raw("s_lshl_b32 %s, %s, 1" % (s(S_HRS + 2), s(S_HRS + 2)))     # hid: M * dff bf16
        st.raw("s_mul_i32 %s, %s, %s" % (s(S_BRS + 2), s(S_NC), s(S_MP)))
        st.raw("s_lshl_b32 %s, %s, 3" % (s(S_BRS + 2), s(S_BRS + 2)))     # mask image: NC * 4 * Mp shorts
    else:
        st.raw("s_mov_b32 %s, 0" % s(S_HRS + 2))
        st.raw("s_mov_b32 %s, 0" % s(S_BRS + 2))
    st.raw("s_lshl_b32 %s, %s, 3" % (s(S_BSTRIDE), s(S_MP)))                # bytes per chunk of the mask image
    st.raw("s_lshr_b32 %s, %s, 2" % (s(S_TMP), s(S_WV)))                    # w
    st.raw("s_mul_i32 %s, %s, %s" % (s(S_BSOFF), s(S_TMP), s(S_MP)))
    st.raw("s_lshl_b32 %s, %s, 2" % (s(S_BSOFF), s(S_BSOFF)))               # + w * 4 * Mp
    st.raw("s_lshl_b32 %s, %s, 7" % (s(S_BI), s(S_TMP)))                    # this wave's 32 units of a chunk: + w * 128 bytes of b1
    st.raw("s_add_u32 %s, %s, %s" % (s(S_B1), s(S_B1), s(S_BI)))
    st.raw("s_addc_u32 %s, %s, 0" % (s(S_B1 + 1), s(S_B1 + 1)))
    st.raw("s_mov_b32 %s, 0" % s(S_BI))
    st.raw("s_lshl_b32 %s, %s, 9" % (s(S_YBASE), s(S_TMP)))                 # the epilogue tile: + w * 512
    st.raw("s_and_b32 %s, %s, 3" % (s(S_TMP), s(S_WV)))                     # pair
    st.raw("s_lshl_b32 %s, %s, 15" % (s(S_TMP), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_YBASE), s(S_YBASE), s(S_TMP)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_YBASE), s(S_YBASE), s(S_SM0)))
    st.raw("s_lshl_b32 %s, %s, 12" % (s(S_W1DST), s(S_WV)))                 # this wave's four 1-KiB pieces of a 32-KiB image
    st.raw("s_add_u32 %s, %s, %s" % (s(S_W1DST), s(S_W1DST), s(S_SM0)))
    st.raw("s_lshl_b32 %s, %s, 8" % (s(S_PBASE), s(S_WV)))
    st.raw("s_add_u32 %s, %s, %s" % (s(S_PBASE), s(S_PBASE), s(S_SM0)))
    st.raw("s_add_u32 %s, %s, 0x10000" % (s(S_PBASE), s(S_PBASE)))
    st.raw("s_mov_b32 %s, 0x80008000" % s(S_K8000))
    st.raw("s_mov_b32 %s, 0x0c0c0301" % s(S_KSEL))
    st.raw("s_sub_u32 %s, %s, 1" % (s(S_TMP), s(S_NC)))
    st.raw("s_lshl_b32 %s, %s, 15" % (s(S_W1LAST), s(S_TMP)))            # (NC - 1) * 32768
    st.raw("s_mov_b32 %s, 0" % s(S_I))
    st.raw("s_min_u32 %s, 0x8000, %s" % (s(S_W1SOFF), s(S_W1LAST)))       # W1 of chunk 1 (chunk 0 again when there is only one)
    st.raw("s_mov_b32 %s, 0" % s(S_W2SOFF))
    st.raw("s_mov_b32 %s, 0" % s(S_HSOFF))
    # ---- the lane's parameters out of LDS (C++ prologue: one dword per thread and parameter, 2 KiB apart) ----
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(WORD))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(WORD), v(WORD)))
    st.raw("v_lshlrev_b32_e32 %s, 2, %s" % (v(WORD), v(WORD)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(WORD), s(S_PBASE), v(WORD)))       # pbase = array + params + wave * 256
    for i, name in enumerate(PARAMS):
        st.raw("ds_read_b32 %s, %s offset:%d" % (v(PARAM_REG[name]), v(WORD), i * 2048))
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("s_barrier")                      # every wave has its parameters: the area (W2 ring) may be overwritten
    # ---- X fragments, W1 of chunk 0 ----
    for ks in range(16):
        st.raw("buffer_load_dwordx4 %s, %s, %s, 0 offen offset:%d" % (v(X + 4 * ks, 4), v(TMP), s(S_XRS, 4), 32 * ks))
    st.raw("s_mov_b32 %s, 0" % s(S_TMP))
    for j in range(4):
        st.raw("s_add_u32 m0, %s, 0x%x" % (s(S_W1DST), j * 1024))
        st.raw("s_nop 0")
        st.raw("buffer_load_dwordx4 %s, %s, %s offen lds" % (v(OFF1 + j), s(S_W1RS, 4), s(S_TMP)))
    # ---- chunk 0's bias -> BI (the scalar block takes over the prologue's temporaries), then chunk 1's on its way ----
    bias_request(st)
    st.raw("s_waitcnt lgkmcnt(0)")
    for j in range(16):
        bias_lo(st, j)
    bias_hi(st)
    bias_advance(st)
    bias_request(st)                                                       # chunk 1 (back by the first iteration's closing wait)
    bias_advance(st)                                                       # next request: chunk 2
    for i in range(NA):
        st.raw("v_accvgpr_write_b32 %s, 0" % a(i))
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    st.raw("s_barrier")
    if (ABL & 64) and TRAIN:
        for k, op in enumerate(("s_memtime", "s_memrealtime")):
            st.raw("%s vcc" % op)
            st.raw("s_waitcnt lgkmcnt(0)")
            st.raw("v_mov_b32_e32 %s, vcc_lo" % v(TMP))
            st.raw("v_accvgpr_write_b32 %s, %s" % (a(NA + k), v(TMP)))
    st.raw("s_cmp_lt_u32 %s, 4" % s(S_WV))
    st.raw("s_cbranch_scc0 .Lrole1_" + U)
    entry = st.snapshot()
    counts0, nops0 = emit_role(st, 0)
    st.label(".Lrole1_" + U)
    st.lds, st.vm, st.mfma_w, st.n = [set(x) for x in entry["lds"]], list(entry["vm"]), dict(entry["mfma"]), entry["n"]
    counts1, nops1 = emit_role(st, 1)
    st.label(".Lend_" + U)
    st.raw("s_waitcnt vmcnt(0) lgkmcnt(0)")
    if (ABL & 64) and TRAIN:
        st.raw("v_accvgpr_read_b32 v0, %s" % a(NA))
        st.raw("v_accvgpr_read_b32 v1, %s" % a(NA + 1))
        for k, op in enumerate(("s_memtime", "s_memrealtime")):
            st.raw("%s vcc" % op)
            st.raw("s_waitcnt lgkmcnt(0)")
            st.raw("v_mov_b32_e32 v%d, vcc_lo" % (2 + k))
        st.raw("buffer_store_dwordx4 v[0:3], %s, %s, 0 offen" % (v(HOFF), s(S_HRS, 4)))
        st.raw("s_waitcnt vmcnt(0)")
    st.raw("s_nop 7")
    st.raw("s_nop 7")
    st.raw("s_barrier")                      # every wave is past its last fragment read: the rings become the epilogue's tiles
    # ---- Y^T to the pair's tile [32 tok][256] f32 (16-byte piece P of token r in slot P ^ (r & 7)): piece 32 w + 8 ytl + 2 g + h ----
    L, R, H, XR, BASE, AD = 0, 1, 2, 3, 4, 5
    st.raw("v_mbcnt_lo_u32_b32 %s, -1, 0" % v(L))
    st.raw("v_mbcnt_hi_u32_b32 %s, -1, %s" % (v(L), v(L)))
    st.raw("v_and_b32_e32 %s, 31, %s" % (v(R), v(L)))
    st.raw("v_lshrrev_b32_e32 %s, 5, %s" % (v(H), v(L)))
    st.raw("v_and_b32_e32 %s, 7, %s" % (v(XR), v(R)))
    st.raw("v_xor_b32_e32 %s, %s, %s" % (v(XR), v(XR), v(H)))
    st.raw("v_lshlrev_b32_e32 %s, 10, %s" % (v(BASE), v(R)))
    st.raw("v_add_u32_e32 %s, %s, %s" % (v(BASE), s(S_YBASE), v(BASE)))
    for g in range(4):
        st.raw("v_xor_b32_e32 %s, %d, %s" % (v(AD + g), 2 * g, v(XR)))
        st.raw("v_lshl_add_u32 %s, %s, 4, %s" % (v(AD + g), v(AD + g), v(BASE)))
    n = 0
    for ytl in range(4):
        for g in range(4):
            t = 16 + 4 * (n & 3)
            n += 1
            for j in range(4):
                st.raw("v_accvgpr_read_b32 %s, %s" % (v(t + j), a(Y + 16 * ytl + 4 * g + j)))
            st.raw("s_nop 0")
            st.raw("ds_write_b128 %s, %s offset:%d" % (v(AD + g), v(t, 4), ytl * 128))
    st.raw("s_waitcnt lgkmcnt(0)")
    st.raw("s_mov_b32 m0, %s" % s(S_M0SAVE))
    return st, counts0, nops0


def main():
    global TRAIN, ABL, DMA_POLICY
    here = os.path.dirname(os.path.abspath(__file__))
    out = os.path.join(os.path.dirname(here), "end-to-end_asr_pytorch_amd", "csrc", "ffn_fwd2_asm.inc")
    args = sys.argv[1:]
    while args and args[0].startswith("--"):
        if args[0] == "--abl":
            ABL = int(args[1])
        elif args[0] == "--policy":
            DMA_POLICY = args[1].replace("+", " ")
        elif args[0] == "--out":
            out = args[1]
        args = args[2:]
    macros = []
    for train in (False, True):
        TRAIN = train
        st, counts, nops = build()
        macros.append(("FFN2_FWD_ASM_%s" % ("TRAIN" if train else "EVAL"), st.out))
        sys.stderr.write("%s: %d lines; per trip of 2 chunks: %s; s_nop states padded in the loop: %d\n" %
                         ("train" if train else "eval", len(st.out), counts, nops))
    regs = ["v%d" % i for i in range(NV)] + ["a%d" % i for i in range(NA)] + ["s%d" % i for i in range(S_LO, S_HI + 1)] + (["a%d" % (NA + k) for k in range(2)] + ["vcc"] if ABL & 64 else []) + ["memory"]
    write_inc(out, "generated by tools/gen_ffn_fwd.py", macros, "FFN2_FWD_ASM_CLOBBERS", regs)
    with open(os.path.join(os.path.dirname(out), "ffn_fwd2_params.h"), "w") as f:
        f.write("// generated by tools/gen_ffn_fwd.py: the order of the per-lane parameters ffn2.hip leaves in LDS for the generated loop\n")
        f.write("enum { %s, FFN2_NPARAMS };\n" % ", ".join("FFN2_P_" + p.upper() for p in PARAMS))


if __name__ == "__main__":
    main()
