// Fused scaled-dot-product attention forward for gfx950 (replaces attention.py:76-84 + the permutes of :47-57).
//
// bf16 path (the product path): flash-style, one workgroup = NW waves x 32 query rows of one (batch, head).
//   * S^T = K.Q^T with v_mfma_f32_32x32x16_bf16 (A = K tile rows from LDS, B = Q rows held in registers), so the
//     query index sits on the LANE and the 32 keys of a half-tile sit in the 16 accumulator registers x 2 lane
//     halves: the online-softmax row max / row sum are 16 in-register ops + one cross-half shuffle, and the
//     running rescale of O is a per-lane scalar.
//   * P^T feeds the PV MFMA directly from those registers as the B operand (O^T = V^T.P^T); the k-order
//     permutation this implies is absorbed by how the A operand (V^T) is read from LDS.
//   * K and V tiles [64 keys][64 d] are copied row-major into a double-buffered, XOR-swizzled LDS image by LDS-DMA
//     (global_load_lds_dwordx4: no staging registers); K fragments are ds_read_b128 rows, V^T fragments come from the same
//     kind of tile through ds_read_b64_tr_b16; the DMA of tile t+1 is in flight during the MFMAs of tile t.
//   * masks come from lengths (key j masked iff j >= k_len[b] or causal && j > i); no mask tensor is read.
//   * scores are BASE-2 logits: q arrives pre-multiplied by log2(e)/sqrt(d_k) (folded into the Q projection's epilogue), so a
//     probability is one v_exp_f32 of an accumulator register - the running maximum is subtracted by the MFMA itself (it is the
//     accumulator's initial value) and is only moved, with the O / row-sum rescale that costs, when a tile's maximum outgrows it
//     by 2^8: the kernel is VALU-bound (one quarter-rate exp per 256 MFMA flops at d_k = 64), every op removed from the
//     per-probability path is throughput.
// fp32 path: one wave per query row, plain VALU, exact fp32 - the parity/debug mode, not a performance path.
#include <stdlib.h>

#include "asr_common.h"

namespace {

// ---------------------------------------------------------------------------------------------------------
// fp32 reference-precision kernel
// ---------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void attn_fwd_f32_kernel(const float* __restrict__ Q, const float* __restrict__ K,
                                                           const float* __restrict__ V, float* __restrict__ ctx,
                                                           float* __restrict__ lse, int B, int h, int Lq, int Lk,
                                                           const int32_t* __restrict__ k_len, int causal) {
    __shared__ float qs[4][64];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int64_t gw = (int64_t)blockIdx.x * 4 + wave;
    if (gw >= (int64_t)B * h * Lq) return;
    const int i = (int)(gw % Lq);
    const int bh = (int)(gw / Lq);
    const int b = bh / h, hd = bh - b * h;
    int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int kend = causal ? min(kl, i + 1) : kl;
    qs[wave][lane] = Q[((int64_t)bh * Lq + i) * 64 + lane];
    __builtin_amdgcn_wave_barrier();
    float m = -INFINITY, l = 0.f, acc = 0.f;
    for (int j0 = 0; j0 < kend; j0 += 64) {
        const int j = j0 + lane;
        const bool valid = j < kend;
        float s = -INFINITY;
        if (valid) {
            const float* kr = K + ((int64_t)bh * Lk + j) * 64;
            float d = 0.f;
#pragma unroll
            for (int dd = 0; dd < 64; dd += 4) {
                const f32x4 kv = *reinterpret_cast<const f32x4*>(kr + dd);
                d = fmaf(qs[wave][dd], kv[0], d);
                d = fmaf(qs[wave][dd + 1], kv[1], d);
                d = fmaf(qs[wave][dd + 2], kv[2], d);
                d = fmaf(qs[wave][dd + 3], kv[3], d);
            }
            s = d;
        }
        const float mc = wave_max(s);
        const float mn = fmaxf(m, mc);
        const float sc = (m == -INFINITY) ? 0.f : exp2f(m - mn);
        const float p = valid ? exp2f(s - mn) : 0.f;
        l = l * sc + wave_sum(p);
        acc *= sc;
        const int cnt = min(64, kend - j0);
        for (int jj = 0; jj < cnt; ++jj) {
            const float pj = __shfl(p, jj, 64);
            acc = fmaf(pj, V[((int64_t)bh * Lk + j0 + jj) * 64 + lane], acc);
        }
        m = mn;
    }
    ctx[((int64_t)b * Lq + i) * (h * 64) + hd * 64 + lane] = acc / l;
    if (lse && lane == 0) lse[(int64_t)bh * Lq + i] = m + log2f(l);   // base-2, like the scores
}

// ---------------------------------------------------------------------------------------------------------
// bf16 MFMA flash kernel
// ---------------------------------------------------------------------------------------------------------
// (timing ablations of these loops - which piece's removal buys how much - are round-3 history: git log, DESIGN / HISTORY.md)
// ---- v2: LDS-DMA staging + hardware-transposed V reads -----------------------------------------------------------------------
// K and V tiles are copied ROW-MAJOR ([64 keys][64 d], 128-byte rows) straight into a double-buffered LDS image by
// global_load_lds_dwordx4 (no staging VGPRs, no register transposes, one barrier per tile, the next tile's DMA in flight during
// the MFMAs).  One swizzle serves both access patterns: 16-byte chunk c of row r lives at slot c ^ f(r),
// f(r) = (((r>>1)&1)<<2) | ((r>>2)&3): conflict-free for the ds_read_b128 row reads of QK^T (16 rows x one chunk) and for the
// ds_read_b64_tr_b16 transposed reads of PV (4 consecutive keys x 64 bytes).  The DMA destination is lane-linear, so the swizzle is
// applied to each lane's SOURCE chunk.  Keys past k_len are clamped to the last valid row (their probabilities are masked to 0).
typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ int swz2(int row) { return (((row >> 1) & 1) << 2) | ((row >> 2) & 3); }

// KS = 2 / 4 (few queries, many keys: the decoder's cross attention, Lq = 51 against Lk = 1000): the NW waves are NW/KS query groups x
// KS key streams - stream kh takes the key tiles t = KS it + kh - and the streams' (reference, row sum, O) states are merged through
// LDS at the end.  One workgroup per (batch, head) then walks 8 / 4 dependent iterations instead of 16: the walk is pure latency
// (128 workgroups of 2 waves on a 256-CU chip), so cutting it cuts the kernel.
// (3 and 4 resident workgroups per CU instead of 2 - 159 / 128 VGPRs - measured within 1 % of this: the loop is not occupancy-bound)
template <int NW, bool CAUSAL, bool DROP, int KS = 1>
__global__ __launch_bounds__(NW * 64, 2) void attn_fwd_bf16_v2_kernel(const bf16_t* __restrict__ Q, const bf16_t* __restrict__ K,
                                                                      const bf16_t* __restrict__ V, bf16_t* __restrict__ ctx,
                                                                      float* __restrict__ lse, int h, int Lq, int Lk,
                                                                      const int32_t* __restrict__ k_len, int q_tiles,
                                                                      asr_dropout_t drop, const uint32_t* __restrict__ drop_bits) {
    constexpr int NWQ = NW / KS, QB = NWQ * 32, PIECES = 8 * KS / NW;   // 1-KiB pieces (8 rows) per wave per operand, per iteration
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 8192 * KS];   // [buf][key stream][K|V]
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave_all = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wave = wave_all % NWQ, kh = wave_all / NWQ;          // query group, key stream
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
    const int q0 = qt * QB;
    const int kl = k_len ? min(k_len[b], Lk) : Lk;
    const int kmax = CAUSAL ? min(kl, q0 + QB) : kl;
    const int ntiles = (kmax + 63) >> 6, niter = (ntiles + KS - 1) / KS;
    const int qrow = q0 + wave * 32 + r;
    const int wave_qlast = q0 + wave * 32 + 31;
    const bf16_t* Kb = K + (int64_t)bh * Lk * 64;
    const bf16_t* Vb = V + (int64_t)bh * Lk * 64;
    // dropout of the probabilities (attention.py:83): keep bits from the Mk image (asr_common.h), this query's column of it
    const int lqp = drop_pad128(Lq);
    const uint32_t* mkp = DROP ? drop_bits + (int64_t)bh * (drop_pad128(Lk) / 32) * lqp + qrow : nullptr;

    u32x4 qf[4];
#pragma unroll
    for (int s = 0; s < 4; ++s)
        qf[s] = (qrow < Lq) ? *reinterpret_cast<const u32x4*>(Q + ((int64_t)bh * Lq + qrow) * 64 + 16 * s + 8 * hh)
                            : u32x4{0, 0, 0, 0};

    // this lane's (row, source chunk) inside each piece it stages: piece p covers tile rows 8p..8p+7
    const int prow = lane >> 3;
    auto stage = [&](int buf, int it) {
#pragma unroll
        for (int i = 0; i < PIECES; ++i) {
            const int pa = wave_all * PIECES + i;            // piece among the 8 * KS of this iteration
            const int p = pa & 7, ks = pa >> 3;
            unsigned char* base = smem + (buf * KS + ks) * 2 * 8192;
            const int key0 = (it * KS + ks) * 64;
            const int row = 8 * p + prow;
            const int c = (lane & 7) ^ swz2(row);
            const int64_t goff = (int64_t)min(key0 + row, kl - 1) * 64 + c * 8;
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Kb + goff),
                                             (__attribute__((address_space(3))) void*)(base + p * 1024), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((const __attribute__((address_space(1))) void*)(Vb + goff),
                                             (__attribute__((address_space(3))) void*)(base + 8192 + p * 1024), 16, 0, 0);
        }
    };

    f32x16 o0, o1;
#pragma unroll
    for (int i = 0; i < 16; ++i) { o0[i] = 0.f; o1[i] = 0.f; }
    // mref: the reference maximum the MFMA subtracts (accumulator init = -mref); the true running maximum never exceeds
    // mref + MAXLAG.  first: nothing accumulated yet (mref is still the placeholder 0).
    constexpr float MAXLAG = 8.f;
    float mref = 0.f, l = 0.f;
    bool first = true;

    if (ntiles > 0) stage(0, 0);
    __syncthreads();
    for (int it = 0; it < niter; ++it) {
        const int cur = it & 1, t = it * KS + kh;
        uint32_t wk[2] = {0u, 0u};
        if (DROP) {   // before the DMA is queued: vmcnt retires in order, queued after it these two words would wait for the whole next tile
            wk[0] = mkp[(int64_t)(2 * t) * lqp] >> (4 * hh);
            wk[1] = mkp[(int64_t)(2 * t + 1) * lqp] >> (4 * hh);
        }
        if (it + 1 < niter) stage(cur ^ 1, it + 1);
        const unsigned char* Ks = smem + (cur * KS + kh) * 2 * 8192;
        const unsigned char* Vs = Ks + 8192;
        const int key0 = t * 64;
        if (t < ntiles && !(CAUSAL && key0 > wave_qlast)) {   // wave-uniform: otherwise the whole tile lies in this wave's future (or past the end)
            f32x16 st[2];
#pragma unroll
            for (int hf = 0; hf < 2; ++hf) {
#pragma unroll
                for (int i = 0; i < 16; ++i) st[hf][i] = -mref;
                const int row = hf * 32 + r;
#pragma unroll
                for (int s = 0; s < 4; ++s) {
                    const u32x4 kf = *reinterpret_cast<const u32x4*>(Ks + row * 128 + (((2 * s + hh) ^ swz2(row)) << 4));
                    st[hf] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, kf),
                                                                     __builtin_bit_cast(bf16x8, qf[s]), st[hf], 0, 0, 0);
                }
            }
            const bool interior = (key0 + 64 <= kl) && (!CAUSAL || key0 + 63 <= q0 + wave * 32);
            float mloc = -INFINITY;
            if (interior) {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int i = 0; i < 16; ++i) mloc = fmaxf(mloc, st[hf][i]);
            } else {
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int i = 0; i < 16; ++i) {
                        const int key = key0 + hf * 32 + (i & 3) + 8 * (i >> 2) + 4 * hh;
                        const bool bad = key >= kl || (CAUSAL && key > qrow);
                        st[hf][i] = bad ? -INFINITY : st[hf][i];
                        mloc = fmaxf(mloc, st[hf][i]);
                    }
            }
            mloc = fmaxf(mloc, __shfl_xor(mloc, 32, 64));     // the other 16 keys of each half-tile live on lane ^ 32
            // move the reference only when it has to: first tile, or this tile's maximum is more than 2^MAXLAG above it
            const bool move = (first && mloc > -INFINITY) || mloc > MAXLAG;
            if (__builtin_amdgcn_ballot_w64(move)) {          // wave-uniform branch; lanes that need no move use delta = 0
                const float delta = move ? mloc : 0.f;
                const float alpha = __builtin_amdgcn_exp2f(-delta);
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[hf][i] -= delta;
                l *= alpha;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] *= alpha; o1[i] *= alpha; }
                mref += delta;
            }
            first = first && !(mloc > -INFINITY);
            f32x2 rs2 = {0.f, 0.f};
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int i = 0; i < 16; i += 2) {
                    const float p0 = __builtin_amdgcn_exp2f(st[hf][i]), p1 = __builtin_amdgcn_exp2f(st[hf][i + 1]);
                    st[hf][i] = p0;
                    st[hf][i + 1] = p1;
                    rs2 += f32x2{p0, p1};
                }
            float rs = rs2[0] + rs2[1];
            rs += __shfl_xor(rs, 32, 64);
            l += rs;
            if (DROP) {   // the row sum above is of the un-dropped probabilities; 1/keep is folded into the final normalisation
#pragma unroll
                for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                    for (int i = 0; i < 16; ++i) st[hf][i] = drop_and(st[hf][i], wk[hf], 8 * (i >> 2) + (i & 3));
            }

            // O^T += V^T . P^T : V^T fragments by transposing reads of the row-major V tile.
            // lane (group g16 = lane>>4, i = lane&15) addresses key row kb + (i>>2), d column dt*32 + 16*(g16&1) + 4*(i&3)
            const int i16 = lane & 15, g16 = lane >> 4;
#pragma unroll
            for (int hf = 0; hf < 2; ++hf)
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    bf16x8 pf;
#pragma unroll
                    for (int j = 0; j < 8; ++j) pf[j] = (bf16_t)st[hf][8 * s2 + j];
                    const int kb = hf * 32 + 16 * s2 + 4 * hh + (i16 >> 2);
#pragma unroll
                    for (int dt = 0; dt < 2; ++dt) {
                        const int col = dt * 32 + 16 * (g16 & 1) + 4 * (i16 & 3);
                        const int c = col >> 3, sub = (col & 7) * 2;
                        const unsigned char* p0 = Vs + kb * 128 + ((c ^ swz2(kb)) << 4) + sub;
                        const unsigned char* p1 = Vs + (kb + 8) * 128 + ((c ^ swz2(kb + 8)) << 4) + sub;
                        const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p0);
                        const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p1);
                        const u32x2 a = __builtin_bit_cast(u32x2, v0), bb = __builtin_bit_cast(u32x2, v1);
                        const u32x4 vf = {a[0], a[1], bb[0], bb[1]};
                        if (dt == 0)
                            o0 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, o0, 0, 0, 0);
                        else
                            o1 = __builtin_amdgcn_mfma_f32_32x32x16_bf16(__builtin_bit_cast(bf16x8, vf), pf, o1, 0, 0, 0);
                    }
                }
        }
        __syncthreads();   // next tile's DMA has landed (barrier fence drains vmcnt) and `cur` may be overwritten
    }

    if (KS > 1) {   // merge the key streams: stream 1 parks its state in LDS (the operand tiles are dead), stream 0 folds it in
        if (kh > 0) {
            float* park = reinterpret_cast<float*>(smem) + (((kh - 1) * NWQ + wave) * 64 + lane) * 36;
            park[0] = mref; park[1] = l; park[2] = first ? 1.f : 0.f;
#pragma unroll
            for (int i = 0; i < 16; ++i) { park[3 + i] = o0[i]; park[19 + i] = o1[i]; }
        }
        __syncthreads();
        if (kh > 0) return;
#pragma unroll
        for (int ks = 1; ks < KS; ++ks) {
            const float* park = reinterpret_cast<const float*>(smem) + (((ks - 1) * NWQ + wave) * 64 + lane) * 36;
            if (park[2] == 0.f) {                 // that stream saw at least one live key
                const float mb = park[0], lb = park[1];
                const float m = first ? mb : fmaxf(mref, mb);
                const float sa = first ? 0.f : __builtin_amdgcn_exp2f(mref - m), sb = __builtin_amdgcn_exp2f(mb - m);
                l = l * sa + lb * sb;
#pragma unroll
                for (int i = 0; i < 16; ++i) { o0[i] = o0[i] * sa + park[3 + i] * sb; o1[i] = o1[i] * sa + park[19 + i] * sb; }
                mref = m;
                first = false;
            }
        }
    }
    if (qrow < Lq) {
        const float inv = (DROP ? drop_scale(drop) : 1.f) / l;
        bf16_t* op = ctx + ((int64_t)b * Lq + qrow) * (h * 64) + hd * 64;
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int d = 8 * g + 4 * hh;
            bf16x4 a = {(bf16_t)(o0[4 * g] * inv), (bf16_t)(o0[4 * g + 1] * inv), (bf16_t)(o0[4 * g + 2] * inv),
                        (bf16_t)(o0[4 * g + 3] * inv)};
            bf16x4 c = {(bf16_t)(o1[4 * g] * inv), (bf16_t)(o1[4 * g + 1] * inv), (bf16_t)(o1[4 * g + 2] * inv),
                        (bf16_t)(o1[4 * g + 3] * inv)};
            *reinterpret_cast<bf16x4*>(op + d) = a;
            *reinterpret_cast<bf16x4*>(op + 32 + d) = c;
        }
        if (lse && hh == 0) lse[(int64_t)bh * Lq + qrow] = mref + __builtin_amdgcn_logf(l);   // base-2
    }
}

// ---- dropout keep bits of one attention call, both images (asr_common.h) ----
// A wave owns 32 queries x 64 keys with the forward kernel's element ownership (lane = query r, half hh; keys hf*32 + 8g + 4hh + x),
// hashes its 16 words, and assembles: its nibbles -> with lane ^ 32 the two full 32-key words of query r (Mk); a 5-step 32 x 32 bit
// transpose across each 32-lane half turns "word of query r" into "word of key c" (Mq).  Both stores are fully coalesced.
struct MaskSites {            // up to 8 dropout sites of one shape hashed by one launch (the decoder's 6 self / 6 cross attention calls)
    asr_dropout_t drop[8];
    uint32_t* bits[8];
};
__global__ __launch_bounds__(256) void attn_dropmask_kernel(MaskSites sites, int BH, int Bn, int h, int Lq, int Lk) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r = lane & 31, hh = lane >> 5;
    const int site = blockIdx.z / BH, bh = blockIdx.z - site * BH, b = bh / h, hd = bh - b * h;
    const asr_dropout_t drop = drop_resolve(sites.drop[site]);
    uint32_t* __restrict__ bits = sites.bits[site];
    const int lqp = drop_pad128(Lq), lkp = drop_pad128(Lk);
    const int qw = blockIdx.y * 4 + wave, qrow = qw * 32 + r, key0 = blockIdx.x * 64;
    if (key0 >= Lk || qw * 32 >= Lq) return;                               // padding only: its bits are unspecified
    const uint32_t dsub = drop_subkey(drop, (uint32_t)(hd * Bn + b));      // leading index head*B + b (attention.py:43-49)
    const uint32_t drow = (uint32_t)qrow * (uint32_t)((Lk + 1) >> 1);
    uint32_t mine[2] = {0u, 0u};
#pragma unroll
    for (int hf = 0; hf < 2; ++hf)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const uint32_t pair = drow + (uint32_t)((key0 + hf * 32 + 8 * g + 4 * hh) >> 1);
            const uint32_t w0 = drop_word(drop, dsub, pair), w1 = drop_word(drop, dsub, pair + 1);
            const uint32_t nib = (drop_keep_lo(drop, w0) ? 1u : 0u) | (drop_keep_hi(drop, w0) ? 2u : 0u) |
                                 (drop_keep_lo(drop, w1) ? 4u : 0u) | (drop_keep_hi(drop, w1) ? 8u : 0u);
            mine[hf] |= nib << (8 * g + 4 * hh);
        }
    const uint32_t full0 = mine[0] | (uint32_t)__shfl_xor((int)mine[0], 32, 64);
    const uint32_t full1 = mine[1] | (uint32_t)__shfl_xor((int)mine[1], 32, 64);
    uint32_t x = hh ? full1 : full0;       // lanes 0..31: keys key0..+31 of query r; lanes 32..63: keys key0+32..+63
    bits[((int64_t)bh * (lkp / 32) + (key0 >> 5) + hh) * lqp + qrow] = x;
    // 32 x 32 bit transpose over the 32 lanes of each half (rows = lanes, columns = bit positions)
#pragma unroll
    for (int j = 16; j >= 1; j >>= 1) {
        const uint32_t m = j == 16 ? 0x0000FFFFu : j == 8 ? 0x00FF00FFu : j == 4 ? 0x0F0F0F0Fu : j == 2 ? 0x33333333u : 0x55555555u;
        const uint32_t y = (uint32_t)__shfl_xor((int)x, j, 64);
        if (r & j) x ^= ((y >> j) ^ x) & m;            // lower-left block <- partner's upper-right
        else       x ^= (((x >> j) ^ y) & m) << j;     // upper-right block <- partner's lower-left
    }
    uint32_t* mq = bits + (int64_t)BH * (lkp / 32) * lqp;
    mq[((int64_t)bh * (lqp / 32) + qw) * lkp + key0 + hh * 32 + r] = x;
}

template <int NW, int KS = 1> int launch_bf16(hipStream_t s, const void* q, const void* k, const void* v, void* ctx, float* lse, int B,
                                  int h, int Lq, int Lk, const int32_t* k_len, int causal, asr_dropout_t drop, const uint32_t* drop_bits) {
    constexpr int QB = NW / KS * 32;
    const int q_tiles = (Lq + QB - 1) / QB;
    dim3 grid(B * h * q_tiles), block(NW * 64);
    {
#define LAUNCH_V2(C, D)                                                                                                        \
    hipLaunchKernelGGL((attn_fwd_bf16_v2_kernel<NW, C, D, KS>), grid, block, 0, s, (const bf16_t*)q, (const bf16_t*)k,         \
                       (const bf16_t*)v, (bf16_t*)ctx, lse, h, Lq, Lk, k_len, q_tiles, drop, drop_bits)
        // the encoder's shape (non-causal, >= 128 queries, one key stream): the generated two-blocks-per-wave kernel (attention_fwd4.hip);
        // it declines (non-zero) what it was not built for and the v2 kernel below takes it
        if (!causal && KS == 1 && NW == 4 && Lq >= 128 &&
            asr_attention_fwd_v4(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, drop, drop_bits) == 0)
            return 0;
        if (causal) { if (drop.thr16) LAUNCH_V2(true, true); else LAUNCH_V2(true, false); }
        else        { if (drop.thr16) LAUNCH_V2(false, true); else LAUNCH_V2(false, false); }
#undef LAUNCH_V2
        ASR_LAUNCH_CHECK("attention_fwd_bf16_v2");
        return 0;
    }
}

}  // namespace


extern "C" int64_t asr_attention_dropmask_words(int B, int h, int Lq, int Lk) { return 2 * drop_mk_words(B * h, Lq, Lk); }

extern "C" int asr_attention_dropmask_multi(void* stream, int n, const asr_dropout_t* drops, uint32_t* const* bits, int B, int h, int Lq,
                                            int Lk) {
    ASR_REQUIRE(drops && bits && n > 0 && n <= 8 && B > 0 && h > 0 && Lq > 0 && Lk > 0, ASR_ERR_ARG, "attention_dropmask: bad args (1..8 sites)");
    ASR_REQUIRE((int64_t)B * h * n <= 65535, ASR_ERR_UNSUPPORTED, "attention_dropmask: B*h*n exceeds the grid's z extent");
    MaskSites sites;
    for (int i = 0; i < 8; ++i) {
        sites.drop[i] = drops[i < n ? i : 0];
        sites.bits[i] = bits[i < n ? i : 0];
    }
    for (int i = 0; i < n; ++i)
        ASR_REQUIRE(bits[i] && drops[i].thr16 > 0 && drops[i].thr16 < 65536u, ASR_ERR_ARG, "attention_dropmask: site %d: null buffer or thr16 outside (0, 65536)", i);
    hipLaunchKernelGGL(attn_dropmask_kernel, dim3(drop_pad128(Lk) / 64, drop_pad128(Lq) / 128, B * h * n), dim3(256), 0,
                       static_cast<hipStream_t>(stream), sites, B * h, B, h, Lq, Lk);
    ASR_LAUNCH_CHECK("attention_dropmask");
    return 0;
}

extern "C" int asr_attention_dropmask(void* stream, asr_dropout_t drop, int B, int h, int Lq, int Lk, uint32_t* bits) {
    return asr_attention_dropmask_multi(stream, 1, &drop, &bits, B, h, Lq, Lk);
}

extern "C" int asr_attention_fwd(void* stream, const void* q, const void* k, const void* v, int dtype, void* ctx, float* lse,
                                 int B, int h, int Lq, int Lk, const int32_t* k_len, int causal, asr_dropout_t drop,
                                 const uint32_t* drop_bits) {
    ASR_REQUIRE(q && k && v && ctx, ASR_ERR_ARG, "attention: null pointer");
    ASR_REQUIRE(!drop.thr16 || drop_bits, ASR_ERR_ARG, "attention: dropout needs the keep-bit images (asr_attention_dropmask)");
    ASR_REQUIRE(drop.thr16 < 65536u, ASR_ERR_ARG, "attention: dropout thr16 must be < 65536");
    ASR_REQUIRE(!(drop.thr16 && dtype != ASR_BF16), ASR_ERR_UNSUPPORTED, "attention: dropout runs on the bf16 (training) path only");
    ASR_REQUIRE(B > 0 && h > 0 && Lq > 0 && Lk > 0, ASR_ERR_ARG, "attention: B=%d h=%d Lq=%d Lk=%d", B, h, Lq, Lk);
    ASR_REQUIRE(asr_aligned(q, 16) && asr_aligned(k, 16) && asr_aligned(v, 16) && asr_aligned(ctx, 16), ASR_ERR_ALIGN,
                "attention: q/k/v/ctx must be 16-byte aligned");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ASR_F32) {
        const int64_t rows = (int64_t)B * h * Lq;
        hipLaunchKernelGGL(attn_fwd_f32_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, s, (const float*)q,
                           (const float*)k, (const float*)v, (float*)ctx, lse, B, h, Lq, Lk, k_len, causal);
        ASR_LAUNCH_CHECK("attention_fwd_f32");
        return 0;
    }
    ASR_REQUIRE(dtype == ASR_BF16, ASR_ERR_ARG, "attention: bad dtype %d", dtype);
    constexpr int xks = 4;      // key streams per workgroup of the few-query kernels (1 / 2: step +0.2 / +0.1 ms)
    if (Lq <= 32) {
        // a handful of queries against a long key sequence (the decode step's cross attention: Lq = 1 or the beam, Lk = the encoder
        // length): one query group, four key streams - a quarter of the dependent walk
        if (xks >= 4 && Lk >= 512 && !causal) return launch_bf16<4, 4>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
        if (xks >= 2 && Lk >= 192 && !causal) return launch_bf16<2, 2>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
        return launch_bf16<1>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
    }
    if (xks >= 4 && Lq <= 64 && Lk >= 512 && !causal) return launch_bf16<8, 4>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);   // cross attention
    if (xks >= 2 && Lq <= 64 && Lk >= 256 && !causal) return launch_bf16<4, 2>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
    if (Lq <= 64) return launch_bf16<2>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
    return launch_bf16<4>(s, q, k, v, ctx, lse, B, h, Lq, Lk, k_len, causal, drop, drop_bits);
}
