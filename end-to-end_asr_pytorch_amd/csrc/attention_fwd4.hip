// Encoder self-attention forward, v4: one wave per SIMD-slot owning 64 query rows as TWO independent 32-row blocks (A, B).
// Replaces attention.py:76-84 (+ the permutes of :47-57) for the non-causal, d_k = d_v = 64, bf16 case.
//
// Why this shape (MI355X_MICROARCH.md 'Per-instruction cycle constants', cdna_hip_programming.md Appendix B 'Fused attention prefill'):
// at d_k = 64 a probability costs 256 MFMA flops - half of what it costs at d_k = 128 - so per 32 x 64 score tile the matrix pipe has
// 16 x 32 = 512 cycles of work while the softmax needs 32 v_exp (8 cycles of issue each) + 32 adds + 16 packs on the SAME SIMD's
// vector issue port.  The loop is bound by vector-instruction ISSUE, not by the matrix pipe; what matters is (1) the number of
// vector instructions per probability and (2) that the matrix pipe never waits for them.
//   (1) No running maximum in the loop.  Scores are base-2 logits (q pre-scaled by log2(e)/sqrt(d_k)); a probability is
//       exp2(s - mref) with mref = 0 until a row's partial sum leaves [2^-64, 2^64] - then, and only then, a wave-uniform rare
//       branch re-centres that wave's rows (exact: a change of reference, O and l are rescaled, the tile's probabilities recomputed
//       from the still-live scores).  bf16 / f32 are floating point: nothing is lost by carrying 2^40 instead of 1.
//       Per probability: 1 v_exp_f32, 1 v_add_f32, 1/2 v_cvt_pk_bf16_f32.
//   (2) Two blocks per wave: while the vector port runs block B's softmax the matrix pipe runs P.V of block A and K.Q^T of A's NEXT
//       tile, and vice versa.  Both streams are independent inside a phase, so the hardware (in order per wave) always has an MFMA
//       to issue between vector instructions.
// Layouts are v3's: S^T = K.Q^T (query on the lane), P^T feeds O^T = V^T.P^T straight from the accumulator registers, K / V tiles
// row-major in a swizzled LDS image filled by LDS-DMA (rings of three 8-KiB slots, requests two tiles ahead, one counted vmcnt and one
// barrier per tile), V^T fragments by ds_read_b64_tr_b16.
#include <stdlib.h>

#include <type_traits>

#include "asr_common.h"

namespace {

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
typedef __attribute__((address_space(3))) unsigned char lds_u8;

__device__ __forceinline__ int swz2(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

__device__ __forceinline__ uint32_t cvt_pk_bf16(float lo, float hi) {
    uint32_t r;
    asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(r) : "v"(lo), "v"(hi));
    return r;
}

#ifndef ATTN4_SCHED
#define ATTN4_SCHED 0
#endif
#ifndef ATTN4_WPE
#define ATTN4_WPE 2          // waves per SIMD the register budget is sized for (2: two workgroups per CU)
#endif

template <bool DROP>
__global__ __launch_bounds__(256, ATTN4_WPE) void attn_fwd_bf16_v4_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                          const bf16_t* __restrict__ V, bf16_t* __restrict__ ctx,
                                                                          float* __restrict__ lse, int h, int Lq, int Lk,
                                                                          const int32_t* __restrict__ k_len, int q_tiles,
                                                                          asr_dropout_t drop, const uint32_t* __restrict__ drop_bits) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[6 * 8192];   // K ring [3][8 KiB] | V ring [3][8 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int qt, bh;
    {
        const int BH = gridDim.x / q_tiles;
        if ((BH & 7) == 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            bh = (slot / q_tiles) * 8 + xcd;
            qt = slot % q_tiles;
        } else {
            qt = blockIdx.x % q_tiles;
            bh = blockIdx.x / q_tiles;
        }
    }
    const int b = bh / h, hd = bh - b * h;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int nt = (kl + 63) >> 6;
    const int qrowA = qt * 256 + wave * 64 + r, qrowB = qrowA + 32;
    const u32x4 krs = rsrc_words(K + (int64_t)bh * Lk * 64, (unsigned)kl * 128u);
    const u32x4 vrs = rsrc_words(V + (int64_t)bh * Lk * 64, (unsigned)kl * 128u);
    const unsigned smem0 = lds_addr_of(smem);

    u32x4 qa[4], qb[4];
    {
        const u32x4 qrs = rsrc_words(Q + (int64_t)bh * Lq * 64, (unsigned)Lq * 128u);
#pragma unroll
        for (int s = 0; s < 4; ++s) {
            qa[s] = load128_asm(qrs, (unsigned)qrowA * 128u + 32u * s + 16u * hh, 0u);
            qb[s] = load128_asm(qrs, (unsigned)qrowB * 128u + 32u * s + 16u * hh, 0u);
        }
    }

    // LDS-DMA: this lane's source offset inside a tile for each of the two 1-KiB pieces its wave stages per operand and tile
    unsigned voff[2];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave * 2 + i) + (lane >> 3);
        voff[i] = (unsigned)(row * 128 + (((lane & 7) ^ swz2(row)) << 4));
    }
    auto stage = [&](const u32x4& rs, int t, unsigned slot_addr) {
        const unsigned soff = t < nt ? (unsigned)t * 8192u : 0x7f000000u;   // past the end: out of range, no fetch (still counted)
#pragma unroll
        for (int i = 0; i < 2; ++i) dma16_asm(rs, voff[i], soff, slot_addr + (wave * 2 + i) * 1024);
    };
    // fragment read offsets inside a tile
    unsigned kofs[4], vofs[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) kofs[s] = (unsigned)(r * 128 + (((2 * s + hh) ^ swz2(r)) << 4));
    {
        const int i16 = lane & 15, g16 = lane >> 4;
        const int kb = 4 * hh + (i16 >> 2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
            const int c = col >> 3, sub = (col & 7) * 2;
            vofs[2 * dt] = (unsigned)(kb * 128 + ((c ^ swz2(kb)) << 4) + sub);
            vofs[2 * dt + 1] = (unsigned)((kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub);
        }
    }

    // ---- the pieces of a phase -------------------------------------------------------------------------------------------------
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // LDS byte addresses of the fragment reads in ring slot 0 (the slot and the sub-tile go into the instructions' offset fields)
    const lds_u8* kbase[4];
    const lds_u8* vbase[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        kbase[i] = (const lds_u8*)(size_t)(smem0 + kofs[i]);
        vbase[i] = (const lds_u8*)(size_t)(smem0 + 3 * 8192 + vofs[i]);
    }
    // S^T = K.Q^T of one block against the K tile at byte offset kso of the K ring
    auto scores = [&](f32x16 (&st)[2], const u32x4 (&qf)[4], int kso) {
        u32x4 kf[2][4];
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                kf[hf][s] = *reinterpret_cast<const __attribute__((address_space(3))) u32x4*>(kbase[s] + kso + hf * 4096);
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
                st[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf[hf][s]), __builtin_bit_cast(bf16x8, qf[s]),
                                                                 s == 0 ? zero16 : st[hf], 0, 0, 0);
    };
    // O^T += V^T.P^T of one block against the V tile at byte offset vso of the V ring
    auto pv = [&](f32x16& o0, f32x16& o1, const u32x4 (&p)[4], int vso) {
#pragma unroll
        for (int g = 0; g < 4; ++g)
#pragma unroll
            for (int dt = 0; dt < 2; ++dt) {
                const u32x2 lo = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vbase[2 * dt] + vso + g * 2048)));
                const u32x2 hi = __builtin_bit_cast(u32x2, __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(vbase[2 * dt + 1] + vso + g * 2048)));
                const u32x4 vf = {lo[0], lo[1], hi[0], hi[1]};
                if (dt == 0) o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, p[g]), o0, 0, 0, 0);
                else o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), __builtin_bit_cast(bf16x8, p[g]), o1, 0, 0, 0);
            }
    };

    bool centred = false;        // wave-uniform: some row of this wave has left reference 0 (the scores then pay one v_sub each)
    // softmax numerators of one block's tile: st (kept intact) -> packed bf16 p; returns this lane's partial row sum
    auto expsum = [&](f32x16 (&st)[2], u32x4 (&p)[4]) -> float {
        float a0 = 0.f, a1 = 0.f;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 16; i += 2) {
                const float p0 = __builtin_amdgcn_exp2f(st[hf][i]), p1 = __builtin_amdgcn_exp2f(st[hf][i + 1]);
                a0 += p0;
                a1 += p1;
                const bf16x2 pk = __builtin_convertvector(f32x2{p0, p1}, bf16x2);
                p[2 * hf + (i >> 3)][(i >> 1) & 3] = __builtin_bit_cast(uint32_t, pk);
            }
        return a0 + a1;
    };
    // the rare branch: re-centre every row of this block (a change of reference: exact up to rounding) and redo the tile's numerators
    auto recentre = [&](f32x16 (&st)[2], u32x4 (&p)[4], float& l, float& mref, f32x16& o0, f32x16& o1) -> float {
        float mloc = -INFINITY;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[hf][i]);
        mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));
        const float lrow = l + __shfl_xor(l, 32, 64);
        float delta = lrow > 0.f ? fmaxf(mloc, __builtin_amdgcn_logf(lrow)) : mloc;
        if (!(delta > -INFINITY)) delta = 0.f;         // nothing live yet in this row (or NaN scores: they stay NaN)
        const float alpha = __builtin_amdgcn_exp2f(-delta);
        l = lrow > 0.f ? 0.5f * lrow * alpha : 0.f;     // the row's sum, shared evenly by its two lanes
        if (lrow > 0.f) {
#pragma unroll
            for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
        }
        mref += delta;
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int i = 0; i < 16; ++i) st[hf][i] -= delta;
        centred = true;
        return expsum(st, p);
    };
    // in range: 2^-64 < x < 2^64 (one unsigned compare on the bit pattern; zero, negatives, inf and NaN all fail)
    auto out_of_range = [&](float x) -> bool { return (__builtin_bit_cast(uint32_t, x) - 0x1f800000u) >= 0x40000000u; };

    // One phase: the vector port works on block X's tile (softmax numerators) while the matrix pipe runs the OTHER block's P.V of the
    // previous tile and K.Q^T of its next one.  CEN / MASK are compile-time so that the common form is ONE basic block and the
    // interleave below can be imposed on it.
    auto phase = [&](auto cen, auto mask, auto sched, bool do_pv, bool do_qk,
                     f32x16& yo0, f32x16& yo1, const u32x4 (&yp)[4], int vso, f32x16 (&ys)[2], const u32x4 (&yq)[4], int kso,
                     f32x16 (&xs)[2], u32x4 (&xp)[4], float& xl, float& xm, f32x16& xo0, f32x16& xo1, int key0) {
        if (do_pv) pv(yo0, yo1, yp, vso);
        if (do_qk) scores(ys, yq, kso);
        if (decltype(cen)::value) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; ++i) xs[hf][i] -= xm;
        }
        if (decltype(mask)::value) {      // the ragged last tile: keys past k_len (their K rows read as zeros) get probability 0
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = key0 + hf * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    xs[hf][i] = key >= kl ? -INFINITY : xs[hf][i];
                }
        }
        float rs = expsum(xs, xp);
        if (decltype(sched)::value) {
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);     // one MFMA
                __builtin_amdgcn_sched_group_barrier(0x100, 2, 0);     // fragment reads of the MFMAs a few gaps on
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);     // two v_exp_f32
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);     // two adds and a pack
                __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x100, 1, 0);
                __builtin_amdgcn_sched_group_barrier(0x400, 2, 0);
                __builtin_amdgcn_sched_group_barrier(0x002, 3, 0);
            }
        }
        if (__builtin_amdgcn_ballot_w64(out_of_range(xl + rs))) rs = recentre(xs, xp, xl, xm, xo0, xo1);
        xl += rs;
    };
    using T = std::true_type;
    using F = std::false_type;
    using SCH = std::integral_constant<bool, (ATTN4_SCHED != 0)>;

    f32x16 oa0 = zero16, oa1 = zero16, ob0 = zero16, ob1 = zero16;
    f32x16 sa[2], sb[2];
    u32x4 pa[4], pb[4];
    float la = 0.f, lb = 0.f, ma = 0.f, mb = 0.f;

    // prologue: K0 V0 K1 | V1 K2 requested; K0 (with Q) waited for; block A's and B's first scores
    stage(krs, 0, smem0);
    stage(vrs, 0, smem0 + 3 * 8192);
    stage(krs, 1, smem0 + 8192);
    stage(vrs, 1, smem0 + 4 * 8192);
    stage(krs, 2, smem0 + 2 * 8192);
    asm volatile("s_waitcnt vmcnt(4)" : "+v"(qa[0]), "+v"(qa[1]), "+v"(qa[2]), "+v"(qa[3]), "+v"(qb[0]), "+v"(qb[1]), "+v"(qb[2]), "+v"(qb[3]) : : "memory");
    __builtin_amdgcn_s_barrier();
    if (nt > 0) {
        scores(sa, qa, 0);
        if (nt == 1) phase(F{}, T{}, F{}, false, true, ob0, ob1, pb, 0, sb, qb, 0, sa, pa, la, ma, oa0, oa1, 0);
        else phase(F{}, F{}, F{}, false, true, ob0, ob1, pb, 0, sb, qb, 0, sa, pa, la, ma, oa0, oa1, 0);
    }
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();       // step 0 requests K(3) into the slot K(0) was read from

    // step t: [softmax B(t) | P.V A(t), K.Q^T A(t+1)]  [softmax A(t+1) | P.V B(t), K.Q^T B(t+1)] ; reads V(t) and K(t+1) only
    // the common form: ring slots static (three steps per trip), no tile of it is the ragged last one, reference 0
    auto step_end = [&]() {
        asm volatile("s_waitcnt vmcnt(4) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
    };
    int t = 0;
#define ATTN4_FAST_STEP(S0, S1, S2)                                                                                              \
    {                                                                                                                             \
        stage(krs, t + 3, smem0 + (S0) * 8192);                                                                                   \
        stage(vrs, t + 2, smem0 + (3 + (S2)) * 8192);                                                                             \
        phase(F{}, F{}, SCH{}, true, true, oa0, oa1, pa, (S0) * 8192, sa, qa, (S1) * 8192, sb, pb, lb, mb, ob0, ob1, 0);            \
        phase(F{}, F{}, SCH{}, true, true, ob0, ob1, pb, (S0) * 8192, sb, qb, (S1) * 8192, sa, pa, la, ma, oa0, oa1, 0);            \
        step_end();                                                                                                               \
        ++t;                                                                                                                      \
    }
    while (t + 4 < nt && !centred) {      // (t + 2 < nt - 2 would do for one step; three are taken per trip)
        ATTN4_FAST_STEP(0, 1, 2)
        ATTN4_FAST_STEP(1, 2, 0)
        ATTN4_FAST_STEP(2, 0, 1)
    }
#undef ATTN4_FAST_STEP
    // the general form: any slot, re-centred rows, the ragged last tile (the last steps of every walk come through here)
    for (; t + 1 < nt; ++t) {
        const int s0 = t % 3, s1 = (t + 1) % 3, s2 = (t + 2) % 3;
        stage(krs, t + 3, smem0 + s0 * 8192);
        stage(vrs, t + 2, smem0 + (3 + s2) * 8192);
        phase(T{}, F{}, F{}, true, true, oa0, oa1, pa, s0 * 8192, sa, qa, s1 * 8192, sb, pb, lb, mb, ob0, ob1, t * 64);
        if (t + 2 == nt) phase(T{}, T{}, F{}, true, true, ob0, ob1, pb, s0 * 8192, sb, qb, s1 * 8192, sa, pa, la, ma, oa0, oa1, (t + 1) * 64);
        else phase(T{}, F{}, F{}, true, true, ob0, ob1, pb, s0 * 8192, sb, qb, s1 * 8192, sa, pa, la, ma, oa0, oa1, (t + 1) * 64);
        step_end();
    }
    if (nt > 0) {
        const int s0 = (nt - 1) % 3;
        phase(T{}, T{}, F{}, true, false, oa0, oa1, pa, s0 * 8192, sa, qa, 0, sb, pb, lb, mb, ob0, ob1, (nt - 1) * 64);
        pv(ob0, ob1, pb, s0 * 8192);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // out-of-range requests of the last steps: nothing may land after the workgroup's LDS is released

    la += __shfl_xor(la, 32, 64);
    lb += __shfl_xor(lb, 32, 64);
    const float dsc = DROP ? drop_scale(drop) : 1.f;
    auto store = [&](const f32x16& o0, const f32x16& o1, float l, float mref, int qrow) {
        if (qrow < Lq) {
            const float inv = dsc / l;
            bf16_t* op = ctx + ((int64_t)b * Lq + qrow) * (h * 64) + hd * 64;
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int d = 8 * g + 4 * hh;
                bf16x4 a = {(bf16_t)(o0[4 * g] * inv), (bf16_t)(o0[4 * g + 1] * inv), (bf16_t)(o0[4 * g + 2] * inv), (bf16_t)(o0[4 * g + 3] * inv)};
                bf16x4 c = {(bf16_t)(o1[4 * g] * inv), (bf16_t)(o1[4 * g + 1] * inv), (bf16_t)(o1[4 * g + 2] * inv), (bf16_t)(o1[4 * g + 3] * inv)};
                *reinterpret_cast<bf16x4*>(op + d) = a;
                *reinterpret_cast<bf16x4*>(op + 32 + d) = c;
            }
            if (lse && hh == 0) lse[(int64_t)bh * Lq + qrow] = mref + __builtin_amdgcn_logf(l);   // base-2
        }
    };
    store(oa0, oa1, la, ma, qrowA);
    store(ob0, ob1, lb, mb, qrowB);
}

// ---- the hand-scheduled form: the whole kernel body is one generated instruction stream (tools/gen_attn_fwd4.py) -------------
// The C++ part only works out this workgroup's pointers and this lane's LDS / global offsets and hands them over in registers
// outside the block's own (v0-v231, a0-a95, s34-s101).
#ifndef ATTN4_INC
#define ATTN4_INC "attention_fwd4_asm.inc"
#endif
#include ATTN4_INC

template <int NW, bool DROP>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_bf16_v4a_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                       const bf16_t* __restrict__ V, bf16_t* __restrict__ ctx,
                                                                       float* __restrict__ lse, int h, int Lq, int Lk,
                                                                       const int32_t* __restrict__ k_len, int q_tiles, float dscale,
                                                                       const uint32_t* __restrict__ drop_bits) {
    constexpr int PIECES = 8 / NW;                      // 1-KiB LDS-DMA pieces per wave, operand and tile
    __shared__ __attribute__((aligned(16))) unsigned char smem[8 * 8192];   // K ring [4][8 KiB] | V ring [4][8 KiB]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int qt, bh;
    {
        const int BH = gridDim.x / q_tiles;
        if ((BH & 7) == 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            bh = (slot / q_tiles) * 8 + xcd;
            qt = slot % q_tiles;
        } else {
            qt = blockIdx.x % q_tiles;
            bh = blockIdx.x / q_tiles;
        }
    }
    const int b = bh / h, hd = bh - b * h;
    const int q0w = qt * (64 * NW) + wave * 64;         // first query row of this wave
    const int qrow = q0w + r;                           // block A's row of this lane; block B's is 32 further
    const unsigned smem0 = lds_addr_of(smem);
    unsigned voff[2], kofs[4], vofs[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {                       // (pieces 2 and 3 of a wave, NW = 2: 16 rows = 2048 source bytes further, same swizzle)
        const int row = 8 * (wave * PIECES + (PIECES > 1 ? i : 0)) + (lane >> 3);
        voff[i] = (unsigned)(row * 128 + (((lane & 7) ^ swz2(row)) << 4));
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) kofs[s] = smem0 + (unsigned)(r * 128 + (((2 * s + hh) ^ swz2(r)) << 4));
    {
        const int i16 = lane & 15, g16 = lane >> 4;
        const int kb = 4 * hh + (i16 >> 2);
#pragma unroll
        for (int dt = 0; dt < 2; ++dt) {
            const int col = dt * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
            const int c = col >> 3, sub = (col & 7) * 2;
            vofs[2 * dt] = smem0 + 32768u + (unsigned)(kb * 128 + ((c ^ swz2(kb)) << 4) + sub);
            vofs[2 * dt + 1] = smem0 + 32768u + (unsigned)((kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub);
        }
    }
    const uint64_t kbp = (uint64_t)(K + (int64_t)bh * Lk * 64), vbp = (uint64_t)(V + (int64_t)bh * Lk * 64);
    const uint64_t qbp = (uint64_t)(Q + (int64_t)bh * Lq * 64);
    const uint64_t cbp = (uint64_t)(ctx + ((int64_t)b * Lq * h + hd) * 64);
    const uint64_t lbp = (uint64_t)(lse ? lse + (int64_t)bh * Lq : nullptr);
    const uint64_t klp = (uint64_t)(k_len ? k_len + b : nullptr);
    const unsigned qoff = (unsigned)qrow * 128u + 16u * hh;
    const unsigned sh4 = 4u * hh;
    // dropout: the Mk keep-bit image of this (batch, head): [key / 32][query] words (asr_common.h)
    const unsigned lqp4 = (unsigned)drop_pad128(Lq) * 4u, msz = DROP ? (unsigned)(drop_pad128(Lk) / 32) * lqp4 : 0u;
    const uint64_t mbp = (uint64_t)(DROP ? drop_bits + (int64_t)bh * (drop_pad128(Lk) / 32) * drop_pad128(Lq) : nullptr);
    const unsigned kdst = smem0 + (unsigned)(wave * PIECES) * 1024u;
    const unsigned stage = smem0 + (unsigned)wave * 9216u;          // the epilogue's 64 rows of 144 bytes
    const unsigned h128 = (unsigned)h * 128u, lse0 = (unsigned)q0w * 4u;
    const unsigned csize = (unsigned)Lq * h128 - (unsigned)hd * 128u;
    const unsigned dsc = __builtin_bit_cast(unsigned, dscale);
#define ATTN4_OPERANDS                                                                                                                     \
    [voff0] "v"(voff[0]), [voff1] "v"(voff[1]), [kofs0] "v"(kofs[0]), [kofs1] "v"(kofs[1]), [kofs2] "v"(kofs[2]), [kofs3] "v"(kofs[3]),    \
        [vofs0] "v"(vofs[0]), [vofs1] "v"(vofs[1]), [vofs2] "v"(vofs[2]), [vofs3] "v"(vofs[3]), [qoff] "v"(qoff), [sh4] "v"(sh4),          \
        [kb] "s"(kbp), [vb] "s"(vbp), [qb] "s"(qbp), [cb] "s"(cbp), [lb] "s"(lbp), [klp] "s"(klp), [lk] "s"(Lk), [kdst] "s"(kdst),         \
        [dsc] "s"(dsc), [h128] "s"(h128), [lq] "s"(Lq), [csize] "s"(csize), [stage] "s"(stage), [lse0] "s"(lse0), [mb] "s"(mbp),           \
        [msz] "s"(msz), [lqp4] "s"(lqp4)
    if constexpr (NW == 4 && DROP) asm volatile(ATTN4_ASM_TRAIN_NW4 : : ATTN4_OPERANDS : ATTN4_ASM_CLOBBERS);
    else if constexpr (NW == 4) asm volatile(ATTN4_ASM_EVAL_NW4 : : ATTN4_OPERANDS : ATTN4_ASM_CLOBBERS);
    else if constexpr (DROP) asm volatile(ATTN4_ASM_TRAIN_NW2 : : ATTN4_OPERANDS : ATTN4_ASM_CLOBBERS);
    else asm volatile(ATTN4_ASM_EVAL_NW2 : : ATTN4_OPERANDS : ATTN4_ASM_CLOBBERS);
#undef ATTN4_OPERANDS
}

}  // namespace

int asr_attention_fwd_v4(hipStream_t s, const void* q, const void* k, const void* v, void* ctx, float* lse, int B, int h, int Lq, int Lk,
                         const int32_t* k_len, asr_dropout_t drop, const uint32_t* drop_bits) {
    static const int form = getenv("ASR_AMD_ATTN_V4") ? atoi(getenv("ASR_AMD_ATTN_V4")) : 2;
    if (form >= 2) {
        // waves per workgroup (64 query rows each): four when that still gives the chip two workgroups per CU, else two
        static const int nw_env = getenv("ASR_AMD_ATTN_NW") ? atoi(getenv("ASR_AMD_ATTN_NW")) : 0;
        const int64_t rows64 = (int64_t)B * h * ((Lq + 63) / 64);
        const int nw = nw_env ? nw_env : (rows64 >= 2048 ? 4 : 2);
        const float dsc = drop.thr16 ? 65536.f / (float)(65536u - drop.thr16) : 1.f;
#define ATTN4_LAUNCH(N, D)                                                                                                              \
    hipLaunchKernelGGL((attn_fwd_bf16_v4a_kernel<N, D>), dim3(B * h * ((Lq + 64 * N - 1) / (64 * N))), dim3(64 * N), 0, s, (const bf16_t*)q, \
                       (const bf16_t*)k, (const bf16_t*)v, (bf16_t*)ctx, lse, h, Lq, Lk, k_len, (Lq + 64 * N - 1) / (64 * N), dsc, drop_bits)
        if (nw >= 4) { if (drop.thr16) ATTN4_LAUNCH(4, true); else ATTN4_LAUNCH(4, false); }
        else         { if (drop.thr16) ATTN4_LAUNCH(2, true); else ATTN4_LAUNCH(2, false); }
#undef ATTN4_LAUNCH
        ASR_LAUNCH_CHECK("attention_fwd_bf16_v4a");
        return 0;
    }
    if (drop.thr16) return -2;
    const int q_tiles = (Lq + 255) / 256;
    hipLaunchKernelGGL((attn_fwd_bf16_v4_kernel<false>), dim3(B * h * q_tiles), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                       (const bf16_t*)v, (bf16_t*)ctx, lse, h, Lq, Lk, k_len, q_tiles, drop, drop_bits);
    ASR_LAUNCH_CHECK("attention_fwd_bf16_v4");
    return 0;
}
