// Flash-attention backward for gfx950 (bf16 MFMA, fp32 accumulate): gradients of attention.py:76-84 without ever
// materialising the [Lq, Lk] probabilities.  P is recomputed from Q, K and the forward's row log-sum-exp.
//
//   Tiles are row-major in LDS, filled by LDS-DMA (double-buffered, one barrier per tile); every transposed operand (K^T, Q^T,
//   dO^T) is fetched with ds_read_b64_tr_b16 from the SAME tile the row reads use (one swizzle serves both, attention.hip v2).
//   kernel A (dQ):   one workgroup = NW waves x 32 queries, query on the LANE (exactly the forward's geometry):
//       S^T = K.Q^T,  dP^T = V.dO^T  (A = K / V tile rows from LDS, B = Q / dO rows in registers),
//       dS^T = P^T o (dP^T - delta[q])   (lse, delta are per-lane scalars),
//       dQ^T += K^T . dS^T               (A = transposed K tile from LDS, B = dS^T straight from the accumulators).
//       Also emits delta[q] = rowsum(dO o O) for kernel B.
//   kernel B (dK,dV): one workgroup = 4 waves x 32 keys, key on the LANE, looping over 64-query tiles:
//       S = Q.K^T, dP = dO.V^T (A = Q / dO tile rows from LDS, B = K / V rows in registers),
//       dV^T += dO^T . P,  dK^T += Q^T . dS   (A = transposed dO / Q tiles from LDS, B = P / dS from the accumulators).
//   Neither kernel needs a cross-workgroup reduction (7 matmuls instead of 5, no float atomics: the dQ atomic
//   traffic of the single-kernel form would cost more than the two recomputed products at L = 1000, d = 64).
// Gradients are written token-major ([B*L, ld] with head h at columns h*64..) so they are directly the A operand of
// the projection GEMMs' backward; dQ carries the 1/sqrt(d_k) that the forward folded into Q.
#include "asr_common.h"

namespace {
constexpr float LN2F = 0.6931471805599453f;
__host__ __device__ __forceinline__ int bwd_pad64(int n) { return (n + 63) & ~63; }

__device__ __forceinline__ bf16x8 pack8(const f32x16& x, int s2) {
    bf16x8 r;
#pragma unroll
    for (int j = 0; j < 8; ++j) r[j] = (bf16_t)x[8 * s2 + j];
    return r;
}
__device__ __forceinline__ f32x16 zero16() {
    f32x16 z;
#pragma unroll
    for (int i = 0; i < 16; ++i) z[i] = 0.f;
    return z;
}
// store a [64 d][32 x] accumulator pair (x on the lane) as token-major bf16 rows: 4 consecutive d per 8-byte store
__device__ __forceinline__ void store_T(bf16_t* rowp, const f32x16& a0, const f32x16& a1, int hh, float scale) {
#pragma unroll
    for (int g = 0; g < 4; ++g) {
        const int d = 8 * g + 4 * hh;
        bf16x4 x = {(bf16_t)(a0[4 * g] * scale), (bf16_t)(a0[4 * g + 1] * scale), (bf16_t)(a0[4 * g + 2] * scale), (bf16_t)(a0[4 * g + 3] * scale)};
        bf16x4 y = {(bf16_t)(a1[4 * g] * scale), (bf16_t)(a1[4 * g + 1] * scale), (bf16_t)(a1[4 * g + 2] * scale), (bf16_t)(a1[4 * g + 3] * scale)};
        *reinterpret_cast<bf16x4*>(rowp + d) = x;
        *reinterpret_cast<bf16x4*>(rowp + 32 + d) = y;
    }
}

// ---- v2 building blocks: row-major tiles by LDS-DMA, one swizzle for row reads and hardware-transposed reads (attention.hip v2)
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int swz2(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }
// DMA one [64 rows][64 bf16] tile: `pieces` 1-KiB pieces (8 rows each) per wave; rows clamped to nrows-1
template <int PIECES>
__device__ __forceinline__ void dma_tile(unsigned char* dst, const bf16_t* src, int64_t ld, int row0, int nrows, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < PIECES; ++i) {
        const int p = wave * PIECES + i;
        const int row = 8 * p + (lane >> 3);
        const int c = (lane & 7) ^ swz2(row);
        __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(src + (int64_t)min(row0 + row, nrows - 1) * ld + c * 8),
                                         (__attribute__((address_space(3))) void*)(dst + p * 1024), 16, 0, 0);
    }
}
__device__ __forceinline__ bf16x8 frag_rows2(const unsigned char* t, int row, int s, int hh) {
    return __builtin_bit_cast(bf16x8, *reinterpret_cast<const u32x4*>(t + row * 128 + (((2 * s + hh) ^ swz2(row)) << 4)));
}
// transposed A-operand fragment (32 columns of the tile x 16 tile-rows in the accumulator's k-order) for column block cb (0/1),
// tile-row block hf (32 rows), k-step s2
__device__ __forceinline__ bf16x8 frag_tr2(const unsigned char* t, int cb, int hf, int s2, int lane) {
    const int i16 = lane & 15, g16 = lane >> 4, hh = lane >> 5;
    const int kb = hf * 32 + 16 * s2 + 4 * hh + (i16 >> 2);
    const int col = cb * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
    const int c = col >> 3, sub = (col & 7) * 2;
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(t + kb * 128 + ((c ^ swz2(kb)) << 4) + sub));
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(t + (kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub));
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}

// ---------------------------------------------------------------------------------------------------------
// KS = 2 / 4: several key streams per workgroup, merged at the end (see attn_fwd_bf16_v2_kernel) - stream kh's NW/KS waves stage and
// consume the key tiles t = KS it + kh.
template <int NW, bool CAUSAL, bool DROP, int KS = 1>
__global__ __launch_bounds__(NW * 64, 2) void attn_bwd_dq_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                 const bf16_t* __restrict__ V, const bf16_t* __restrict__ O,
                                                                 const bf16_t* __restrict__ dO, const float* __restrict__ lse,
                                                                 float* __restrict__ delta, bf16_t* __restrict__ dq_out, int64_t ldq,
                                                                 int h, int Lq, int Lk, const int32_t* __restrict__ k_len, int q_tiles,
                                                                 float scale, asr_dropout_t drop, const uint32_t* __restrict__ drop_bits) {
    constexpr int NWQ = NW / KS, QB = NWQ * 32, PIECES = 8 / NWQ;
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 8192 * KS];   // [buf][key stream][K|V], row-major, LDS-DMA filled
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % NWQ, kh = wave_all / NWQ;          // query group, key stream
    int qt, bh;   // XCD-aware map (see attention.hip): tiles of one (batch, head) share an XCD and are adjacent in time
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
    const int q0 = qt * QB;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int kmax = CAUSAL ? min(kl, q0 + QB) : kl;
    const int ntiles = (kmax + 63) >> 6, niter = (ntiles + KS - 1) / KS;
    const int qrow = q0 + wave * 32 + r;
    const int wave_qlast = q0 + wave * 32 + 31;
    const bool qok = qrow < Lq;
    const bf16_t* Kb = K + (int64_t)bh * Lk * 64;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 64;
    const int64_t tok = ((int64_t)b * Lq + qrow) * (h * 64) + hd * 64;  // token-major row of this lane's query
    const int lqp = drop_pad128(Lq);
    const uint32_t* mkp = DROP ? drop_bits + (int64_t)bh * (drop_pad128(Lk) / 32) * lqp + qrow : nullptr;   // Mk image (asr_common.h)
    const float dsc = DROP ? drop_scale(drop) : 1.f;

    bf16x8 qf[4], dof[4];
    float dl = 0.f;
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 16 * s + 8 * hh;
        const u32x4 z = {0, 0, 0, 0};
        qf[s] = __builtin_bit_cast(bf16x8, qok ? *reinterpret_cast<const u32x4*>(Q + ((int64_t)bh * Lq + qrow) * 64 + c) : z);
        dof[s] = __builtin_bit_cast(bf16x8, qok ? *reinterpret_cast<const u32x4*>(dO + tok + c) : z);
        const bf16x8 of = __builtin_bit_cast(bf16x8, qok ? *reinterpret_cast<const u32x4*>(O + tok + c) : z);
#pragma unroll
        for (int j = 0; j < 8; ++j) dl += (float)dof[s][j] * (float)of[j];
    }
    dl += __shfl_xor(dl, 32, 64);
    const float my_lse2 = qok ? lse[(int64_t)bh * Lq + qrow] : 0.f;   // base-2, like the scores (q carries log2(e)/sqrt(d_k))
    // the per-row scalars the dK / dV kernel starts its accumulators from, NEGATED and padded to whole 64-query tiles (asr_hip.h):
    // [0] -delta (0 in the padding), [1] -lse (-inf in the padding: a padded query's probability is exactly 0)
    {
        const int lqp = bwd_pad64(Lq);
        float* nd = delta + (int64_t)bh * lqp;
        float* nl = delta + ((int64_t)(gridDim.x / q_tiles) + bh) * lqp;
        if (hh == 0 && kh == 0 && qrow < lqp) {
            nd[qrow] = qok ? -dl : 0.f;
            nl[qrow] = qok ? -my_lse2 : -INFINITY;
        }
        if (hh == 0 && kh == 0 && qt == q_tiles - 1)         // (a grid of 32-row tiles can end short of the 64-row padding)
            for (int qq = qrow + QB; qq < lqp; qq += QB) { nd[qq] = 0.f; nl[qq] = -INFINITY; }
    }

    f32x16 a0 = zero16(), a1 = zero16();
    f32x16 neglse, zeros = zero16();
#pragma unroll
    for (int i = 0; i < 16; ++i) neglse[i] = -my_lse2;
    asm volatile("" : "+v"(neglse), "+v"(zeros));      // kept in registers: rematerialised they are the 64 v_mov again
    auto stage = [&](int buf, int it) {      // each key stream's waves stage their own tile
        const int t = it * KS + kh;
        dma_tile<PIECES>(smem + (buf * KS + kh) * 16384, Kb, 64, t * 64, kl, wave, lane);
        dma_tile<PIECES>(smem + (buf * KS + kh) * 16384 + 8192, Vb, 64, t * 64, kl, wave, lane);
    };
    if (ntiles > 0) stage(0, 0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int t = it * KS + kh, key0 = t * 64, cur = it & 1;
        uint32_t wk[2] = {0u, 0u};
        if (DROP) {   // ahead of the DMA: vmcnt retires in order, queued behind it these two words would wait for the whole next tile
            wk[0] = mkp[(int64_t)(2 * t) * lqp] >> (4 * hh);
            wk[1] = mkp[(int64_t)(2 * t + 1) * lqp] >> (4 * hh);
        }
        if (it + 1 < niter) stage(cur ^ 1, it + 1);
        const unsigned char* Ks = smem + (cur * KS + kh) * 16384;
        const unsigned char* Vs = Ks + 8192;
        if (t >= ntiles || (CAUSAL && key0 > wave_qlast)) { __syncthreads(); continue; }
        // interior tile: every key valid for every (in-range) query of this wave -> no mask arithmetic (wave-uniform)
        const bool interior = (key0 + 64 <= kl) && (!CAUSAL || key0 + 63 <= q0 + wave * 32) && (q0 + wave * 32 + 31 < Lq);
        f32x16 st[2], dp[2];
        // the score accumulators start at -lse of this lane's query (-inf for a masked key on an edge tile): exp2 of the finished
        // product is the probability, no subtraction and no select per element.  On interior tiles that start value is the C operand
        // of the first MFMA of each chain - a persistent register set (neglse, zero16) instead of 64 v_mov per tile in a VALU-bound loop
        if (!interior) {
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; ++i) {
                    const int key = key0 + hf * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                    const bool bad = key >= kl || (CAUSAL && key > qrow) || !qok;
                    st[hf][i] = bad ? -INFINITY : -my_lse2;
                }
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            const int row = hf * 32 + r;
            if (interior) {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    st[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(Ks, row, s, hh), qf[s], s == 0 ? neglse : st[hf], 0, 0, 0);
                    dp[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(Vs, row, s, hh), dof[s], s == 0 ? zeros : dp[hf], 0, 0, 0);
                }
            } else {
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    st[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(Ks, row, s, hh), qf[s], st[hf], 0, 0, 0);
                    dp[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(Vs, row, s, hh), dof[s], s == 0 ? zeros : dp[hf], 0, 0, 0);
                }
            }
            if (DROP) {   // dP = dropout mask * (dO . V^T); the 1/keep scale rides on the fma below
#pragma unroll
                for (int i = 0; i < 16; ++i) dp[hf][i] = drop_and(dp[hf][i], wk[hf], 8 * (i >> 2) + (i & 3));
            }
#pragma unroll
            for (int i = 0; i < 16; ++i) {
                const float p = __builtin_amdgcn_exp2f(st[hf][i]);
                st[hf][i] = p * (DROP ? __builtin_fmaf(dp[hf][i], dsc, -dl) : dp[hf][i] - dl);  // dS^T
            }
        }
#pragma unroll
        for (int hf = 0; hf < 2; ++hf)
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(st[hf], s2);
                a0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(Ks, 0, hf, s2, lane), pf, a0, 0, 0, 0);   // K^T from the same K tile
                a1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(Ks, 1, hf, s2, lane), pf, a1, 0, 0, 0);
            }
        __syncthreads();   // next tile's DMA landed; `cur` may be overwritten
    }
    if (KS > 1) {   // fold key stream 1's partial dQ into stream 0's
        if (kh > 0) {
            float* park = reinterpret_cast<float*>(smem) + (((kh - 1) * NWQ + wave) * 64 + lane) * 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) { park[i] = a0[i]; park[16 + i] = a1[i]; }
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
            const float* park = reinterpret_cast<const float*>(smem) + (((ks - 1) * NWQ + wave) * 64 + lane) * 32;
#pragma unroll
            for (int i = 0; i < 16; ++i) { a0[i] += park[i]; a1[i] += park[16 + i]; }
        }
    }
    if (qok) store_T(dq_out + ((int64_t)b * Lq + qrow) * ldq + hd * 64, a0, a1, hh, scale);
}

// ---------------------------------------------------------------------------------------------------------
template <bool CAUSAL, bool DROP>
__global__ __launch_bounds__(256, 2) void attn_bwd_dkv_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                              const bf16_t* __restrict__ V, const bf16_t* __restrict__ dO,
                                                              const float* __restrict__ lse, const float* __restrict__ delta,
                                                              bf16_t* __restrict__ dk_out, bf16_t* __restrict__ dv_out, int64_t ldkv,
                                                              int h, int Lq, int Lk, const int32_t* __restrict__ k_len, int k_tiles,
                                                              asr_dropout_t drop, const uint32_t* __restrict__ drop_bits) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 8192 + 1024];   // [buf][Q|dO] row-major + [buf][lse|delta]
    const int tid = threadIdx.x, lane = tid & 63, r = lane & 31, hh = lane >> 5;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    int kt, bh;   // XCD-aware map: key blocks of one (batch, head) share an XCD (they all stream the same Q / dO)
    {
        const int BH = gridDim.x / k_tiles;
        if ((BH & 7) == 0) {
            const int xcd = blockIdx.x & 7, slot = blockIdx.x >> 3;
            bh = (slot / k_tiles) * 8 + xcd;
            kt = slot % k_tiles;
        } else {
            kt = blockIdx.x % k_tiles;
            bh = blockIdx.x / k_tiles;
        }
    }
    const int b = bh / h, hd = bh - b * h;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int key = kt * 128 + wave * 32 + r;
    const bool kok = key < kl;
    const bf16_t* Qb = Q + (int64_t)bh * Lq * 64;
    const bf16_t* dOb = dO + (int64_t)b * Lq * (h * 64) + hd * 64;   // token-major rows, ld = h*64
    const int64_t ldo = (int64_t)h * 64;
    // Mq image (asr_common.h): after the Mk image; this key's column
    const int lkp = drop_pad128(Lk), lqp = drop_pad128(Lq);
    const uint32_t* mqp = DROP ? drop_bits + drop_mk_words(gridDim.x / k_tiles, Lq, Lk) + (int64_t)bh * (lqp / 32) * lkp + key : nullptr;
    const float dsc = DROP ? drop_scale(drop) : 1.f;

    bf16x8 kf[4], vf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s) {
        const int c = 16 * s + 8 * hh;
        const u32x4 z = {0, 0, 0, 0};
        kf[s] = __builtin_bit_cast(bf16x8, kok ? *reinterpret_cast<const u32x4*>(K + ((int64_t)bh * Lk + key) * 64 + c) : z);
        vf[s] = __builtin_bit_cast(bf16x8, kok ? *reinterpret_cast<const u32x4*>(V + ((int64_t)bh * Lk + key) * 64 + c) : z);
    }
    f32x16 dk0 = zero16(), dk1 = zero16(), dv0 = zero16(), dv1 = zero16();
    const int qt_first = CAUSAL ? (kt * 128) / 64 : 0;   // queries before the first key of this block see none of it
    const int nqt = (Lq + 63) >> 6;
    const int wave_kfirst = kt * 128 + wave * 32;
    auto stage = [&](int buf, int t) {
        const int q0 = t * 64;
        dma_tile<2>(smem + buf * 16384, Qb, 64, q0, Lq, wave, lane);
        dma_tile<2>(smem + buf * 16384 + 8192, dOb, ldo, q0, Lq, wave, lane);
        // per-row scalars of the tile: waves 0 / 1 DMA 64 floats of -lse / -delta (4 bytes per lane; padded to whole tiles by the dQ kernel)
        float* sc = reinterpret_cast<float*>(smem + 32768) + buf * 128;
        const int lqp = bwd_pad64(Lq);
        const int q = q0 + lane;
        if (wave == 0)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(delta + ((int64_t)(gridDim.x / k_tiles) + bh) * lqp + q),
                                             (__attribute__((address_space(3))) void*)sc, 4, 0, 0);
        else if (wave == 1)
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(delta + (int64_t)bh * lqp + q),
                                             (__attribute__((address_space(3))) void*)(sc + 64), 4, 0, 0);
    };
    if (qt_first < nqt) stage(0, qt_first);
    __syncthreads();
    for (int t = qt_first; t < nqt; ++t) {
        const int q0 = t * 64, cur = (t - qt_first) & 1;
        uint32_t wq[2] = {0u, 0u};
        if (DROP) {   // ahead of the DMA (in-order vmcnt)
            wq[0] = mqp[(int64_t)(2 * t) * lkp] >> (4 * hh);
            wq[1] = mqp[(int64_t)(2 * t + 1) * lkp] >> (4 * hh);
        }
        if (t + 1 < nqt) stage(cur ^ 1, t + 1);
        const unsigned char* Qs = smem + cur * 16384;
        const unsigned char* dOs = Qs + 8192;
        const float* lse_s = reinterpret_cast<const float*>(smem + 32768) + cur * 128;
        const float* del_s = lse_s + 64;
        if (CAUSAL && q0 + 63 < wave_kfirst) { __syncthreads(); continue; }   // every query of the tile precedes every key of this wave
        // interior tile: all 64 queries in range, all 32 keys of the wave valid and (causal) not in any query's future
        const bool interior = (q0 + 64 <= Lq) && (wave_kfirst + 31 < kl) && (!CAUSAL || wave_kfirst + 31 <= q0);
        f32x16 sq[2], dp[2];
        auto out_products = [&](int hf) {
#pragma unroll
            for (int s2 = 0; s2 < 2; ++s2) {
                const bf16x8 pf = pack8(sq[hf], s2), sf = pack8(dp[hf], s2);
                dv0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(dOs, 0, hf, s2, lane), pf, dv0, 0, 0, 0);   // dO^T from the dO tile
                dv1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(dOs, 1, hf, s2, lane), pf, dv1, 0, 0, 0);
                dk0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(Qs, 0, hf, s2, lane), sf, dk0, 0, 0, 0);    // Q^T from the Q tile
                dk1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_tr2(Qs, 1, hf, s2, lane), sf, dk1, 0, 0, 0);
            }
        };
#pragma unroll
        for (int hf = 0; hf < 2; ++hf) {
            // the score accumulator starts at -lse of its query row (base-2), so exp2 of the finished product IS the probability; on
            // an edge tile a masked (key, query) pair starts at -inf instead and comes out as exactly 0 - the interior tiles carry no
            // mask arithmetic at all (the select per element cost 5 VALU instructions of the 15 per element this loop had).  (Both
            // halves initialised under one branch ahead of the loop - what the dQ kernel does - measured 8 % slower here.)
            if (interior) {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + hf * 32 + 8 * g + 4 * hh);   // base-2
#pragma unroll
                    for (int i = 0; i < 4; ++i) sq[hf][4 * g + i] = l4[i];
                }
            } else {
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int ql = hf * 32 + 8 * g + 4 * hh;
                    const f32x4 l4 = *reinterpret_cast<const f32x4*>(lse_s + ql);
#pragma unroll
                    for (int i = 0; i < 4; ++i) {
                        const int q = q0 + ql + i;
                        const bool bad = !kok || q >= Lq || (CAUSAL && key > q);
                        sq[hf][4 * g + i] = bad ? -INFINITY : l4[i];
                    }
                }
            }
            dp[hf] = zero16();
            const int row = hf * 32 + r;
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                sq[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(Qs, row, s, hh), kf[s], sq[hf], 0, 0, 0);
                dp[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag_rows2(dOs, row, s, hh), vf[s], dp[hf], 0, 0, 0);
            }
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int ql = hf * 32 + 8 * g + 4 * hh;   // 4 consecutive query rows live in regs 4g..4g+3
                const f32x4 d4 = *reinterpret_cast<const f32x4*>(del_s + ql);
#pragma unroll
                for (int i = 0; i < 4; ++i) {
                    const float p = __builtin_amdgcn_exp2f(sq[hf][4 * g + i]);
                    if (DROP) {
                        sq[hf][4 * g + i] = drop_and(p, wq[hf], 8 * g + i);              // the kept P (dropout's 1 / keep factor is applied to dV once, at the store)
                        dp[hf][4 * g + i] = p * __builtin_fmaf(drop_and(dp[hf][4 * g + i], wq[hf], 8 * g + i), dsc, d4[i]);     // dS
                    } else {
                        sq[hf][4 * g + i] = p;
                        dp[hf][4 * g + i] = p * (dp[hf][4 * g + i] + d4[i]);
                    }
                }
            }
        }
        out_products(0);
        out_products(1);
        __syncthreads();   // next tile's DMA landed; `cur` may be overwritten
    }
    if (key < Lk) {   // keys in [kl, Lk) get exact zeros
        const int64_t off = ((int64_t)b * Lk + key) * ldkv + hd * 64;
        store_T(dk_out + off, dk0, dk1, hh, 0.6931471805599453f);   // dS is per natural-log score; q carries log2(e): dK = dS^T.q * ln 2
        store_T(dv_out + off, dv0, dv1, hh, dsc);
    }
}


// ---------------------------------------------------------------------------------------------------------
// fp32 parity-mode backward (the partner of attn_fwd_f32_kernel): one wavefront per (batch, head, query), lane = head dimension for
// the row vectors and = key for the scores; dK / dV are accumulated with float atomics into zeroed buffers.  Not a fast kernel: it
// exists so that the backward tape can run end to end in fp32 and be compared with the reference's gradients at 1e-4 instead of
// through bf16 rounding.
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_bwd_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K, const float* __restrict__ V,
                                                           const float* __restrict__ O, const float* __restrict__ dO,
                                                           const float* __restrict__ lse, float* __restrict__ dq, int64_t ldq,
                                                           float* __restrict__ dk, float* __restrict__ dv, int64_t ldkv, int B, int h, int Lq,
                                                           int Lk, const int32_t* __restrict__ k_len, int causal, float scale) {
    __shared__ float qs[4][64], gs[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    if (gw >= (int64_t)B * h * Lq) return;
    const int i = (int)(gw % Lq);
    const int bh = (int)(gw / Lq);
    const int b = bh / h, hd = bh - b * h;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int kend = causal ? min(kl, i + 1) : kl;
    const float qd = Q[((int64_t)bh * Lq + i) * 64 + lane];                         // scaled query (scale * log2(e) folded in)
    const float gd = dO[((int64_t)b * Lq + i) * (h * 64) + hd * 64 + lane];
    const float od = O[((int64_t)b * Lq + i) * (h * 64) + hd * 64 + lane];
    qs[wave][lane] = qd;
    gs[wave][lane] = gd;
    __builtin_amdgcn_wave_barrier();
    const float delta = wave_sum(gd * od);
    const float l2 = lse[(int64_t)bh * Lq + i];
    float dq_acc = 0.f;
    for (int j0 = 0; j0 < kend; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < kend;
        float p = 0.f, ds = 0.f;
        if (valid) {
            const float* kr = K + ((int64_t)bh * Lk + j) * 64;
            const float* vr = V + ((int64_t)bh * Lk + j) * 64;
            float s = 0.f, dp = 0.f;
#pragma unroll 8
            for (int dd = 0; dd < 64; ++dd) {
                s = fmaf(qs[wave][dd], kr[dd], s);
                dp = fmaf(gs[wave][dd], vr[dd], dp);
            }
            p = exp2f(s - l2);
            ds = p * (dp - delta);
        }
        const int cnt = min(64, kend - j0);
        for (int jj = 0; jj < cnt; ++jj) {
            const float pj = __shfl(p, jj, 64), dsj = __shfl(ds, jj, 64);
            const int64_t krow = (int64_t)bh * Lk + j0 + jj, orow = ((int64_t)b * Lk + j0 + jj) * ldkv + hd * 64 + lane;
            dq_acc = fmaf(dsj, K[krow * 64 + lane], dq_acc);
            atomicAdd(dk + orow, dsj * qd * LN2F);                                  // q carries log2(e): d/dk of the natural-log score
            atomicAdd(dv + orow, pj * gd);
        }
    }
    dq[((int64_t)b * Lq + i) * ldq + hd * 64 + lane] = dq_acc * scale;              // gradient wrt the unscaled query, times scale
}

}  // namespace

extern "C" int64_t asr_attention_bwd_workspace_floats(int B, int h, int Lq) { return 2 * (int64_t)B * h * bwd_pad64(Lq); }

extern "C" int asr_attention_bwd_dq(void* stream, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                    const float* lse, float* delta, void* dq, int64_t ldq, int B, int h, int Lq, int Lk,
                                    const int32_t* k_len, int causal, float scale, asr_dropout_t drop, const uint32_t* drop_bits) {
    ASR_REQUIRE(q && k && v && o && d_o && lse && delta && dq, ASR_ERR_ARG, "attention_bwd_dq: null pointer");
    ASR_REQUIRE(!drop.thr16 || drop_bits, ASR_ERR_ARG, "attention_bwd_dq: dropout needs the keep-bit images (asr_attention_dropmask)");
    ASR_REQUIRE(drop.thr16 < 65536u, ASR_ERR_ARG, "attention_bwd_dq: dropout thr16 must be < 65536");
    ASR_REQUIRE(B > 0 && h > 0 && Lq > 0 && Lk > 0, ASR_ERR_ARG, "attention_bwd_dq: bad sizes");
    ASR_REQUIRE(asr_aligned(q, 16) && asr_aligned(k, 16) && asr_aligned(v, 16) && asr_aligned(o, 16) && asr_aligned(d_o, 16) &&
                    asr_aligned(dq, 8) && ldq % 4 == 0, ASR_ERR_ALIGN, "attention_bwd_dq: alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (!causal && asr_attention_bwd_dq_v4(s, q, k, v, o, d_o, lse, delta, dq, ldq, B, h, Lq, Lk, k_len, scale, drop, drop_bits) == 0) return 0;
    const bf16_t *Q = (const bf16_t*)q, *K = (const bf16_t*)k, *V = (const bf16_t*)v, *O = (const bf16_t*)o, *dO = (const bf16_t*)d_o;
#define LAUNCH_DQ2(NW, C, D, KS)                                                                                          \
    hipLaunchKernelGGL((attn_bwd_dq_kernel<NW, C, D, KS>), dim3(B * h * q_tiles), dim3(NW * 64), 0, s, Q, K, V, O, dO, lse, delta, \
                       (bf16_t*)dq, ldq, h, Lq, Lk, k_len, q_tiles, scale, drop, drop_bits)
#define LAUNCH_DQ(NW, KS)                                                                                    \
    do {                                                                                                     \
        const int q_tiles = (Lq + NW / KS * 32 - 1) / (NW / KS * 32);                                        \
        if (causal) { if (drop.thr16) LAUNCH_DQ2(NW, true, true, KS); else LAUNCH_DQ2(NW, true, false, KS); }  \
        else { if (drop.thr16) LAUNCH_DQ2(NW, false, true, KS); else LAUNCH_DQ2(NW, false, false, KS); }       \
    } while (0)
    constexpr int xks = 4;
    if (Lq <= 32) LAUNCH_DQ(1, 1);
    else if (xks >= 4 && Lq <= 64 && Lk >= 512 && !causal) LAUNCH_DQ(8, 4);      // the decoder's cross attention: several key streams per workgroup
    else if (xks >= 2 && Lq <= 64 && Lk >= 256 && !causal) LAUNCH_DQ(4, 2);
    else if (Lq <= 64) LAUNCH_DQ(2, 1);
    else LAUNCH_DQ(4, 1);
#undef LAUNCH_DQ
#undef LAUNCH_DQ2
    ASR_LAUNCH_CHECK("attention_bwd_dq");
    return 0;
}

extern "C" int asr_attention_bwd_dkv(void* stream, const void* q, const void* k, const void* v, const void* d_o, const float* lse,
                                     const float* delta, void* dk, void* dv, int64_t ldkv, int B, int h, int Lq, int Lk,
                                     const int32_t* k_len, int causal, asr_dropout_t drop, const uint32_t* drop_bits) {
    ASR_REQUIRE(q && k && v && d_o && lse && delta && dk && dv, ASR_ERR_ARG, "attention_bwd_dkv: null pointer");
    ASR_REQUIRE(!drop.thr16 || drop_bits, ASR_ERR_ARG, "attention_bwd_dkv: dropout needs the keep-bit images (asr_attention_dropmask)");
    ASR_REQUIRE(drop.thr16 < 65536u, ASR_ERR_ARG, "attention_bwd_dkv: dropout thr16 must be < 65536");
    ASR_REQUIRE(B > 0 && h > 0 && Lq > 0 && Lk > 0, ASR_ERR_ARG, "attention_bwd_dkv: bad sizes");
    ASR_REQUIRE(asr_aligned(q, 16) && asr_aligned(k, 16) && asr_aligned(v, 16) && asr_aligned(d_o, 16) && asr_aligned(dk, 8) &&
                    asr_aligned(dv, 8) && ldkv % 4 == 0, ASR_ERR_ALIGN, "attention_bwd_dkv: alignment");
    hipStream_t s = static_cast<hipStream_t>(stream);
    const bf16_t *Q = (const bf16_t*)q, *K = (const bf16_t*)k, *V = (const bf16_t*)v, *dO = (const bf16_t*)d_o;
    if (!causal && asr_attention_bwd_dkv_v4(s, q, k, v, d_o, delta, dk, dv, ldkv, B, h, Lq, Lk, k_len, drop, drop_bits) == 0) return 0;
    const int k_tiles = (Lk + 127) / 128;
#define LAUNCH_DKV(C, D)                                                                                                         \
    hipLaunchKernelGGL((attn_bwd_dkv_kernel<C, D>), dim3(B * h * k_tiles), dim3(256), 0, s, Q, K, V, dO, lse, delta, (bf16_t*)dk, \
                       (bf16_t*)dv, ldkv, h, Lq, Lk, k_len, k_tiles, drop, drop_bits)
    if (causal) { if (drop.thr16) LAUNCH_DKV(true, true); else LAUNCH_DKV(true, false); }
    else { if (drop.thr16) LAUNCH_DKV(false, true); else LAUNCH_DKV(false, false); }
#undef LAUNCH_DKV
    ASR_LAUNCH_CHECK("attention_bwd_dkv");
    return 0;
}

extern "C" int asr_attention_bwd(void* stream, const void* q, const void* k, const void* v, const void* o, const void* d_o,
                                 const float* lse, float* delta, void* dq, int64_t ldq, void* dk, void* dv, int64_t ldkv, int B, int h,
                                 int Lq, int Lk, const int32_t* k_len, int causal, float scale, asr_dropout_t drop,
                                 const uint32_t* drop_bits) {
    if (int rc = asr_attention_bwd_dq(stream, q, k, v, o, d_o, lse, delta, dq, ldq, B, h, Lq, Lk, k_len, causal, scale, drop, drop_bits))
        return rc;
    return asr_attention_bwd_dkv(stream, q, k, v, d_o, lse, delta, dk, dv, ldkv, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
}

extern "C" int asr_attention_bwd_f32(void* stream, const float* q, const float* k, const float* v, const float* o, const float* d_o,
                                     const float* lse, float* dq, int64_t ldq, float* dk, float* dv, int64_t ldkv, int B, int h, int Lq,
                                     int Lk, const int32_t* k_len, int causal, float scale) {
    ASR_REQUIRE(q && k && v && o && d_o && lse && dq && dk && dv, ASR_ERR_ARG, "attention_bwd_f32: null pointer");
    ASR_REQUIRE(B > 0 && h > 0 && Lq > 0 && Lk > 0 && ldq >= h * 64 && ldkv >= h * 64, ASR_ERR_ARG, "attention_bwd_f32: bad sizes");
    hipStream_t s = static_cast<hipStream_t>(stream);
    // dk / dv are accumulated: zero the [B*Lk, h*64] windows (row stride ldkv) first
    hipError_t e = hipMemset2DAsync(dk, (size_t)ldkv * 4, 0, (size_t)h * 64 * 4, (size_t)B * Lk, s);
    if (e == hipSuccess) e = hipMemset2DAsync(dv, (size_t)ldkv * 4, 0, (size_t)h * 64 * 4, (size_t)B * Lk, s);
    if (e != hipSuccess) { asr_set_error("attention_bwd_f32 memset: %s", hipGetErrorString(e)); return (int)e; }
    const int64_t waves = (int64_t)B * h * Lq;
    hipLaunchKernelGGL(attn_bwd_f32_kernel, dim3((unsigned)((waves + 3) / 4)), dim3(256), 0, s, q, k, v, o, d_o, lse, dq, ldq, dk, dv, ldkv,
                       B, h, Lq, Lk, k_len, causal, scale);
    ASR_LAUNCH_CHECK("attention_bwd_f32");
    return 0;
}
