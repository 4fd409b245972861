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
    {
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
}
