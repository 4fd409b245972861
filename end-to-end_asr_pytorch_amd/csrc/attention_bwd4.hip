// Attention backward, dK / dV half, hand-scheduled (the instruction stream is generated: tools/gen_attn_bwd.py ->
// attention_bwd_asm.inc; see that script's header for the structure).  Differentiates attention.py:76-84 for the non-causal,
// d_k = d_v = 64, bf16 case with Lq >= 128 (the encoder's self attention); everything else stays on attention_bwd.hip's kernels.
// The C++ part only works out this workgroup's pointers and this lane's LDS / global offsets.
#include <stdlib.h>

#include "asr_common.h"

namespace {

__device__ __forceinline__ int swz2(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
__host__ __device__ __forceinline__ int bwd_pad64(int n) { return (n + 63) & ~63; }

#include "attention_bwd_asm.inc"

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_v4_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                                 const float* __restrict__ nscal, bf16_t* __restrict__ dk_out,
                                                                 bf16_t* __restrict__ dv_out, int64_t ldkv, int h, int Lq, int Lk,
                                                                 const int32_t* __restrict__ k_len, int k_tiles, float dscale,
                                                                 const uint32_t* __restrict__ drop_bits) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 512 + 4 * 16384];   // [4 x (-lse 256 | -delta 256)][4 x (Q tile 8192 | dO tile 8192)]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int kt, bh;   // XCD-aware map: key blocks of one (batch, head) share an XCD (they all stream the same Q / dO)
    const int BH = gridDim.x / k_tiles;
    if ((BH & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / k_tiles) * 8 + xcd;
        kt = slot % k_tiles;
    } else {
        kt = blockIdx.x % k_tiles;
        bh = blockIdx.x / k_tiles;
    }
    const int b = bh / h, hd = bh - b * h;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int key0 = kt * 128 + wave * 32;
    const unsigned smem0 = lds_addr_of(smem);
    const unsigned h128 = (unsigned)h * 128u;
    unsigned voffq[2], voffd[2], rb[4], tb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave * 2 + i) + (lane >> 3);
        const unsigned chunk = (unsigned)(((lane & 7) ^ swz2(row)) << 4);
        voffq[i] = (unsigned)row * 128u + chunk;
        voffd[i] = (unsigned)row * h128 + chunk;
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rb[s] = smem0 + 2048u + (unsigned)(r * 128 + (((2 * s + hh) ^ swz2(r)) << 4));
    {
        const int i16 = lane & 15, g16 = lane >> 4;
        const int kb = 4 * hh + (i16 >> 2);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = cb * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
            const int c = col >> 3, sub = (col & 7) * 2;
            tb[2 * cb] = smem0 + 2048u + (unsigned)(kb * 128 + ((c ^ swz2(kb)) << 4) + sub);
            tb[2 * cb + 1] = smem0 + 2048u + (unsigned)((kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub);
        }
    }
    const unsigned vq = voffq[0] | (voffq[1] << 16), rb01 = rb[0] | (rb[1] << 16), rb23 = rb[2] | (rb[3] << 16);
    const unsigned tb01 = tb[0] | (tb[1] << 16), tb23 = tb[2] | (tb[3] << 16);
    const unsigned kfoff = (unsigned)(key0 + r) * 128u + 16u * hh;
    const int lqp = bwd_pad64(Lq);
    const uint64_t qbp = (uint64_t)(Q + (int64_t)bh * Lq * 64), dobp = (uint64_t)(dO + ((int64_t)b * Lq * h + hd) * 64);
    const uint64_t kbp = (uint64_t)(K + (int64_t)bh * Lk * 64), vbp = (uint64_t)(V + (int64_t)bh * Lk * 64);
    // even waves request the -lse run of a tile, odd waves the -delta run (asr_attention_bwd_workspace_floats: [0] -delta, [1] -lse)
    const uint64_t smbp = (uint64_t)(nscal + ((wave & 1) ? (int64_t)bh : (int64_t)BH + bh) * lqp);
    const uint64_t dkbp = (uint64_t)(dk_out + (int64_t)b * Lk * ldkv + hd * 64), dvbp = (uint64_t)(dv_out + (int64_t)b * Lk * ldkv + hd * 64);
    const unsigned ldkv2 = (unsigned)ldkv * 2u;
    const unsigned dsc = __builtin_bit_cast(unsigned, dscale);
    const int lkp = drop_pad128(Lk), lqp128 = drop_pad128(Lq);
    const uint64_t mqbp = (uint64_t)(DROP ? drop_bits + drop_mk_words(BH, Lq, Lk) + (int64_t)bh * (lqp128 / 32) * lkp : nullptr);
    const unsigned msz = DROP ? (unsigned)(lqp128 / 32) * (unsigned)lkp * 4u : 0u, lkp4 = (unsigned)lkp * 4u;
#define ATTN_BWD_OPERANDS                                                                                                               \
    [voffq] "v"(vq), [voffd0] "v"(voffd[0]), [voffd1] "v"(voffd[1]), [rb01] "v"(rb01), [rb23] "v"(rb23), [tb01] "v"(tb01),               \
        [tb23] "v"(tb23), [kfoff] "v"(kfoff), [qb] "s"(qbp), [dob] "s"(dobp), [kb] "s"(kbp), [vb] "s"(vbp), [smb] "s"(smbp),             \
        [dkb] "s"(dkbp), [dvb] "s"(dvbp), [kl] "s"(kl), [lq] "s"(Lq), [lk] "s"(Lk), [h128] "s"(h128), [ldkv2] "s"(ldkv2), [dsc] "s"(dsc), \
        [key0] "s"(key0), [wave] "s"(wave), [smem0] "s"(smem0), [mqb] "s"(mqbp), [msz] "s"(msz), [lkp4] "s"(lkp4)
    if constexpr (DROP) asm volatile(ATTN_BWD_DKV_ASM_TRAIN : : ATTN_BWD_OPERANDS : ATTN_BWD_ASM_CLOBBERS);
    else asm volatile(ATTN_BWD_DKV_ASM_EVAL : : ATTN_BWD_OPERANDS : ATTN_BWD_ASM_CLOBBERS);
#undef ATTN_BWD_OPERANDS
}

#include "attention_bwd_dq_asm.inc"

template <bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dq_v4_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                const bf16_t* __restrict__ V, const bf16_t* __restrict__ O,
                                                                const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                                float* __restrict__ nscal, bf16_t* __restrict__ dq_out, int64_t ldq, int h,
                                                                int Lq, int Lk, const int32_t* __restrict__ k_len, int q_tiles, float scale,
                                                                float dscale, const uint32_t* __restrict__ drop_bits) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 16384];   // ring of 4 slots: K tile 8192 | V tile 8192
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, hh = lane >> 5;
    int qt, bh;
    const int BH = gridDim.x / q_tiles;
    if ((BH & 7) == 0) {
        const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
        bh = (slot / q_tiles) * 8 + xcd;
        qt = slot % q_tiles;
    } else {
        qt = blockIdx.x % q_tiles;
        bh = blockIdx.x / q_tiles;
    }
    const int b = bh / h, hd = bh - b * h;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int row0 = qt * 128 + wave * 32;
    const unsigned smem0 = lds_addr_of(smem);
    const unsigned h128 = (unsigned)h * 128u;
    unsigned voff[2], rb[4], tb[4];
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int row = 8 * (wave * 2 + i) + (lane >> 3);
        voff[i] = (unsigned)row * 128u + (unsigned)(((lane & 7) ^ swz2(row)) << 4);
    }
#pragma unroll
    for (int s = 0; s < 4; ++s) rb[s] = smem0 + (unsigned)(r * 128 + (((2 * s + hh) ^ swz2(r)) << 4));
    {
        const int i16 = lane & 15, g16 = lane >> 4;
        const int kb = 4 * hh + (i16 >> 2);
#pragma unroll
        for (int cb = 0; cb < 2; ++cb) {
            const int col = cb * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
            const int c = col >> 3, sub = (col & 7) * 2;
            tb[2 * cb] = smem0 + (unsigned)(kb * 128 + ((c ^ swz2(kb)) << 4) + sub);
            tb[2 * cb + 1] = smem0 + (unsigned)((kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub);
        }
    }
    const unsigned vo = voff[0] | (voff[1] << 16), rb01 = rb[0] | (rb[1] << 16), rb23 = rb[2] | (rb[3] << 16);
    const unsigned tb01 = tb[0] | (tb[1] << 16), tb23 = tb[2] | (tb[3] << 16);
    const unsigned qoff = (unsigned)(row0 + r) * 128u + 16u * hh, tokoff = (unsigned)(row0 + r) * h128 + 16u * hh;
    const int lqp = bwd_pad64(Lq);
    const uint64_t kbp = (uint64_t)(K + (int64_t)bh * Lk * 64), vbp = (uint64_t)(V + (int64_t)bh * Lk * 64);
    const uint64_t qbp = (uint64_t)(Q + (int64_t)bh * Lq * 64);
    const uint64_t dobp = (uint64_t)(dO + ((int64_t)b * Lq * h + hd) * 64), obp = (uint64_t)(O + ((int64_t)b * Lq * h + hd) * 64);
    const uint64_t lbp = (uint64_t)(lse + (int64_t)bh * Lq), wbp = (uint64_t)(nscal + (int64_t)bh * lqp);
    const uint64_t dqbp = (uint64_t)(dq_out + (int64_t)b * Lq * ldq + hd * 64);
    const unsigned nloff = (unsigned)BH * (unsigned)lqp * 4u, ldq2 = (unsigned)ldq * 2u;
    const unsigned dsc = __builtin_bit_cast(unsigned, dscale), scl = __builtin_bit_cast(unsigned, scale);
    const int lqp128 = drop_pad128(Lq);
    const uint64_t mkbp = (uint64_t)(DROP ? drop_bits + (int64_t)bh * (drop_pad128(Lk) / 32) * lqp128 : nullptr);
    const unsigned msz = DROP ? (unsigned)(drop_pad128(Lk) / 32) * (unsigned)lqp128 * 4u : 0u, lqp4 = (unsigned)lqp128 * 4u;
#define ATTN_BWD_DQ_OPERANDS                                                                                                             \
    [voff] "v"(vo), [rb01] "v"(rb01), [rb23] "v"(rb23), [tb01] "v"(tb01), [tb23] "v"(tb23), [qoff] "v"(qoff), [tokoff] "v"(tokoff),        \
        [kb] "s"(kbp), [vb] "s"(vbp), [qb] "s"(qbp), [dob] "s"(dobp), [ob] "s"(obp), [lb] "s"(lbp), [wb] "s"(wbp), [dqb] "s"(dqbp),        \
        [kl] "s"(kl), [lq] "s"(Lq), [h128] "s"(h128), [ldq2] "s"(ldq2), [dsc] "s"(dsc), [scale] "s"(scl), [row0] "s"(row0),                \
        [nloff] "s"(nloff), [wave] "s"(wave), [smem0] "s"(smem0), [mkb] "s"(mkbp), [msz] "s"(msz), [lqp4] "s"(lqp4)
    if constexpr (DROP) asm volatile(ATTN_BWD_DQ_ASM_TRAIN : : ATTN_BWD_DQ_OPERANDS : ATTN_BWD_DQ_ASM_CLOBBERS);
    else asm volatile(ATTN_BWD_DQ_ASM_EVAL : : ATTN_BWD_DQ_OPERANDS : ATTN_BWD_DQ_ASM_CLOBBERS);
#undef ATTN_BWD_DQ_OPERANDS
}

}  // namespace

// 0 = launched, -2 = not this kernel's case.  nscal: the backward's workspace (asr_attention_bwd_workspace_floats), filled here
int asr_attention_bwd_dq_v4(hipStream_t s, const void* q, const void* k, const void* v, const void* o, const void* d_o, const float* lse,
                            float* nscal, void* dq, int64_t ldq, int B, int h, int Lq, int Lk, const int32_t* k_len, float scale,
                            asr_dropout_t drop, const uint32_t* drop_bits) {
    constexpr int on = 3;      // bit 0: dK / dV, bit 1: dQ on the generated streams
    if (!(on & 2) || Lq < 128) return -2;
    const int q_tiles = (Lq + 127) / 128;
    const float dsc = drop.thr16 ? 65536.f / (float)(65536u - drop.thr16) : 1.f;
    if (drop.thr16)
        hipLaunchKernelGGL((attn_bwd_dq_v4_kernel<true>), dim3(B * h * q_tiles), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o, lse, nscal, (bf16_t*)dq, ldq, h, Lq, Lk, k_len, q_tiles,
                           scale, dsc, drop_bits);
    else
        hipLaunchKernelGGL((attn_bwd_dq_v4_kernel<false>), dim3(B * h * q_tiles), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, (const bf16_t*)o, (const bf16_t*)d_o, lse, nscal, (bf16_t*)dq, ldq, h, Lq, Lk, k_len, q_tiles,
                           scale, dsc, drop_bits);
    ASR_LAUNCH_CHECK("attention_bwd_dq_v4");
    return 0;
}

// 0 = launched, -2 = not this kernel's case.  nscal: the workspace the dQ kernel filled (-delta, -lse; padded to whole 64-query tiles)
int asr_attention_bwd_dkv_v4(hipStream_t s, const void* q, const void* k, const void* v, const void* d_o, const float* nscal, void* dk,
                             void* dv, int64_t ldkv, int B, int h, int Lq, int Lk, const int32_t* k_len, asr_dropout_t drop,
                             const uint32_t* drop_bits) {
    constexpr int on = 3;      // bit 0: dK / dV, bit 1: dQ on the generated streams
    if (!(on & 1) || Lq < 128) return -2;
    const int k_tiles = (Lk + 127) / 128;
    const float dsc = drop.thr16 ? 65536.f / (float)(65536u - drop.thr16) : 1.f;
    if (drop.thr16)
        hipLaunchKernelGGL((attn_bwd_dkv_v4_kernel<true>), dim3(B * h * k_tiles), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, (const bf16_t*)d_o, nscal, (bf16_t*)dk, (bf16_t*)dv, ldkv, h, Lq, Lk, k_len, k_tiles, dsc, drop_bits);
    else
        hipLaunchKernelGGL((attn_bwd_dkv_v4_kernel<false>), dim3(B * h * k_tiles), dim3(256), 0, s, (const bf16_t*)q, (const bf16_t*)k,
                           (const bf16_t*)v, (const bf16_t*)d_o, nscal, (bf16_t*)dk, (bf16_t*)dv, ldkv, h, Lq, Lk, k_len, k_tiles, dsc, drop_bits);
    ASR_LAUNCH_CHECK("attention_bwd_dkv_v4");
    return 0;
}
