// Row-block kernels of the encoder layer at d_model = 256 (a workgroup owns 128 tokens, four waves x 32, one wave per SIMD; the tokens'
// 256-wide rows sit in REGISTERS as MFMA B operands, the weights stream through 32-KiB LDS images by LDS-DMA):
//   asr_ffn_bwd / asr_ffn_bwd_ln   the feed-forward sub-layer's data gradient in one launch   dH = (ds W2) * relu'(H),  dX = dH W1 + ds
//                                  (autograd of src/transformer/module.py:48-53); per 64-unit chunk dH^T = (W2c^T . ds^T) * mask and
//                                  dX^T += W1c^T . dH^T, dH written once (bf16, for the weight gradient dW1 = dH^T X)
//   asr_proj_ln_fwd                attention.py:58-60: fc -> dropout -> + residual -> layer_norm
//   asr_proj_heads (rows form)     attention.py:43-49: the head-major Q / K / V projections
//   asr_ffn_fwd                    argument checks only - the forward launch is ffn2.hip (two waves per SIMD, generated loop); the mask image
//                                  it leaves (uint16 [chunk][tile t][lane half h][token]) is read here by the data gradient
// In every kernel the [M, d_ff] or [M, 3 d] intermediate never makes a round trip through HBM inside the launch, and rows leave as
// whole 512-byte / 1-KiB lines through a per-wave LDS tile.
#include "asr_common.h"

#include <stdlib.h>
#include <type_traits>

namespace {

#ifndef FFN_HID_POLICY
#define FFN_HID_POLICY 0      // cache policy of the hidden-gradient stores (A/B in round 3: 2 = non-temporal: the step unchanged)
#endif
constexpr int FBM = 128;      // tokens per workgroup (4 waves x 32)
constexpr int FHC = 64;       // hidden units per chunk
constexpr int FD = 256;       // d_model
constexpr int W1BUF = FHC * FD * 2;   // 32 KiB: [64 hidden][256 k] bf16, 512-byte rows
constexpr int W2BUF = FD * FHC * 2;   // 32 KiB: [256 d][64 hidden] bf16, 128-byte rows
constexpr int FFN_MAX_DFF = 2048;
constexpr int HST_BYTES = 4 * 4096;   // per wave: a chunk's H^T tile (32 tokens x 128 B) on its way to full-line stores

typedef __attribute__((address_space(3))) void lds_void;

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

// sum over the 64 lanes, the same value in every lane: four DPP adds inside the 16-lane rows, then the four row sums by v_readlane
template <int CTRL> __device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_perm<0xB1>(v);      // quad_perm [1,0,3,2]
    v += dpp_perm<0x4E>(v);      // quad_perm [2,3,0,1]
    v += dpp_perm<0x141>(v);     // row_half_mirror
    v += dpp_perm<0x140>(v);     // row_mirror
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

struct ProjLnArgs {
    const bf16_t* x16;      // the attention context [M, 256] (h * d_v)
    const float* x32;       // the residual: the layer's input
    const bf16_t* w1;       // fc.weight [256][256]
    const float* b2;        // fc.bias
    const float* gamma;
    const float* beta;
    const int32_t* row_len;
    float* s_out;
    float* y32;
    bf16_t* y16;
    float* mean;
    float* rstd;
    int M, L;
    float eps;
    asr_dropout_t drop;
};

// The attention sub-layer's tail at encoder size (attention.py:58-60: fc -> dropout -> + residual -> layer_norm): one product on the row-block
// structure - a workgroup owns 128 tokens (4 waves x 32, rows in registers as MFMA B operands), the [256][256] output projection streams
// through two 32-KiB LDS images as four 64-row chunks whose accumulators ARE the 256 outputs of a token, the epilogue is bias + dropout +
// residual + LayerNorm + row mask over complete rows.  (The feed-forward sub-layer's forward, which this kernel was cut from, is ffn2.hip.)
template <bool TRAIN, bool DROP>
__global__ __launch_bounds__(256, 1) void proj_ln_kernel(const ProjLnArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[4 * 32768];
    unsigned char* const w1s = smem;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * FBM + wave * 32 + r;
    const int mc = m < a.M ? m : a.M - 1;

    // ---- weight staging: a buffer descriptor over the matrix, per-lane byte offsets fixed for the kernel's life --------------------------
    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w1), 0, FD * FD * 2, 0x00020000);
    unsigned off1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // piece p: chunk rows 2p, 2p + 1 (512 B each); LDS slot pc of row u holds 16-byte chunk (pc & 16) | ((pc ^ u) & 15)
        const int p = wave * 8 + k, u = 2 * p + (lane >> 5), pc = lane & 31;
        off1[k] = (unsigned)(u * FD * 2 + ((pc & 16) | ((pc ^ u) & 15)) * 16);
    }
    auto dma_w1 = [&](int buf, int chunk, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(w1s + buf * W1BUF + (wave * 8 + j) * 1024), 16, off1[j], chunk * (FHC * FD * 2), 0, 0);
    };
    const int u15 = r & 15;      // (the accumulator rows are the outputs themselves, in natural order)
    unsigned a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = (unsigned)(r * 512 + (((2 * kk + h) ^ u15) << 4));

    // ---- this lane's token as the product's B operand: ctx[m][16 ks + 8 h .. + 8] -----------------------------------------------------
    bf16x8 xb[16];
    {
        const bf16_t* xr = a.x16 + (int64_t)mc * FD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_w1(0, 0, j);
    f32x16 Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[t][j] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    f32x4 res[32];      // the epilogue's residual rows (row layout: lane = 4 columns of a token row), requested beside the last chunk
    const int m0 = blockIdx.x * FBM + wave * 32;
    auto frag1 = [&](const unsigned char* w1, int k) {      // MFMA k: k-step k >> 1, row tile k & 1
        const int ks = k >> 1, t = k & 1;
        return *reinterpret_cast<const bf16x8*>(w1 + a1[ks & 7] + t * 16384 + (ks >> 3) * 256);
    };
    // the row-mask bits (a load of row_len[b] per row in the epilogue was a dependent global load + wait per row, ~1000 cycles each)
    uint32_t keepmask = 0xffffffffu;
    const int m0c = m0 < a.M ? m0 : a.M - 1;
    const int b_first = m0c / a.L, t_first = m0c - b_first * a.L;      // (one division; rows advance from here)
    if (a.row_len) {
        keepmask = 0;
        int bb = b_first, tt = t_first, len = a.row_len[bb];
        for (int tr = 0; tr < 32; ++tr) {
            keepmask |= (tt < len ? 1u : 0u) << tr;
            if (++tt == a.L) {
                tt = 0;
                if (m0 + tr + 1 < a.M) len = a.row_len[++bb];
            }
        }
    }
    // four chunks of the projection through the two images: 128 MFMAs, 2 us of a launch that is its epilogue
    auto chunk = [&](auto C) {
        constexpr int c = decltype(C)::value;
        const unsigned char* w1 = w1s + (c & 1) * W1BUF;
        if constexpr (c + 1 < 4) {
#pragma unroll
            for (int j = 0; j < 8; ++j) dma_w1((c + 1) & 1, c + 1, j);
        } else {
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                const int rowc = m0 + k < a.M ? m0 + k : a.M - 1;
                res[k] = *reinterpret_cast<const f32x4*>(a.x32 + (int64_t)rowc * FD + 4 * lane);
            }
        }
#pragma unroll
        for (int k = 0; k < 32; ++k)
            Y[2 * c + (k & 1)] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(frag1(w1, k), xb[k >> 1], Y[2 * c + (k & 1)], 0, 0, 0);
        asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    chunk(std::integral_constant<int, 0>{});
    chunk(std::integral_constant<int, 1>{});
    chunk(std::integral_constant<int, 2>{});
    chunk(std::integral_constant<int, 3>{});

    // ---- epilogue: v = dropout(Y + b2) + x, LayerNorm, row mask --------------------------------------------------------------------
    // A lane holds 128 values of ITS token (stores from here would touch 32 rows per instruction, 32 bytes each).  The weight images
    // are free now: each wave parks its Y^T tile in 32 KiB of them ([32 tokens][256] f32, 16-byte piece p of token r in slot
    // p ^ (r & 7): conflict-free both ways) and reads it back one token row per instruction - the three stores are then whole
    // 1-KiB / 512-byte rows, and the row arithmetic is add_layernorm_fwd_kernel's (lane = 4 columns of the row).
    __syncthreads();
    unsigned char* const tile = smem + wave * 32768;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(tile + r * 1024 + (((8 * t + 2 * g + h) ^ (r & 7)) << 4)) =
                f32x4{Y[t][4 * g], Y[t][4 * g + 1], Y[t][4 * g + 2], Y[t][4 * g + 3]};
    const asr_dropout_t drop = drop_resolve(a.drop);
    const float sc = drop_scale(drop);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(a.b2 + 4 * lane);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(a.gamma + 4 * lane), bt = *reinterpret_cast<const f32x4*>(a.beta + 4 * lane);
    const auto rss = __builtin_amdgcn_make_buffer_rsrc(a.s_out, 0, a.s_out ? (int)((int64_t)a.M * FD * 4) : 0, 0x00020000);
    const auto rsy = __builtin_amdgcn_make_buffer_rsrc(a.y32, 0, (int)((int64_t)a.M * FD * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(a.y16, 0, a.y16 ? (int)((int64_t)a.M * FD * 2) : 0, 0x00020000);
    const auto rsm = __builtin_amdgcn_make_buffer_rsrc(a.mean, 0, a.mean ? a.M * 4 : 0, 0x00020000);
    const auto rsr = __builtin_amdgcn_make_buffer_rsrc(a.rstd, 0, a.rstd ? a.M * 4 : 0, 0x00020000);
    int bb = b_first, tt = t_first;              // (utterance, position) of the row being hashed: uniform, advanced row by row
    uint32_t sub = DROP ? drop_subkey(drop, (uint32_t)bb) : 0u;
    float mean_l = 0.f, rstd_l = 0.f;            // lane tr keeps row tr's statistics: one 128-byte store each at the end
#pragma unroll
    for (int tr0 = 0; tr0 < 32; tr0 += 8) {
        f32x4 v[8];
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {      // eight rows at a time: their reductions interleave
            const int tr = tr0 + j;
            const f32x4 y = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (j & 7)) << 4));
            f32x4 w = y + b2v;
            if (DROP) {
                w = drop4(drop, sub, (uint32_t)tt, FD >> 1, (uint32_t)(4 * lane), w, sc);
                if (++tt == a.L) {
                    tt = 0;
                    sub = drop_subkey(drop, (uint32_t)++bb);
                }
            }
            v[j] = w + res[tr];
            part[j] = (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
        float mean[8], rstd[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) mean[j] = wave_sum_dpp(part[j]) * (1.f / FD);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const f32x4 dl = v[j] - mean[j];
            part[j] = (dl[0] * dl[0] + dl[1] * dl[1]) + (dl[2] * dl[2] + dl[3] * dl[3]);
        }
#pragma unroll
        for (int j = 0; j < 8; ++j) rstd[j] = 1.0f / sqrtf(wave_sum_dpp(part[j]) * (1.f / FD) + a.eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tr = tr0 + j, row = m0 + tr;
            const bool rv = row < a.M;
            const unsigned o16 = rv ? (unsigned)row * (FD * 4u) + 16u * lane : 0x80000000u;
            if (TRAIN) {
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v[j]), rss, o16, 0, 0);
                mean_l = lane == tr ? mean[j] : mean_l;
                rstd_l = lane == tr ? rstd[j] : rstd_l;
            }
            f32x4 o = (v[j] - mean[j]) * rstd[j] * gm + bt;
            if (!((keepmask >> tr) & 1u)) o = f32x4{0.f, 0.f, 0.f, 0.f};
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rsy, o16, 0, 0);
            const bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rsz, rv ? (unsigned)row * (FD * 2u) + 8u * lane : 0x80000000u, 0, 0);
        }
    }
    if (TRAIN) {
        const unsigned o4 = (lane < 32 && m0 + lane < a.M) ? (unsigned)(m0 + lane) * 4u : 0x80000000u;
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, mean_l), rsm, o4, 0, 0);
        __builtin_amdgcn_raw_buffer_store_b32(__builtin_bit_cast(uint32_t, rstd_l), rsr, o4, 0, 0);
    }
}

// ---- data gradient: dH^T = (W2c^T . ds^T) * mask, dX^T += W1c^T . dH^T -----------------------------------------------------------
// Same decomposition, same (lane, register) <-> (token, hidden unit) map as the forward, so a lane reads back exactly the mask words it
// wrote.  Both products now sum over the index the weights are NOT contiguous in (W2 [256][d_ff] over its rows, W1 [d_ff][256] over its
// rows), so the weight chunks are staged row-major as stored and every A fragment is two ds_read_b64_tr_b16 (4 k-rows x 16 columns per
// 16-lane group, delivered column-major).  The column quads a group's four address lanes point at are free: quads 1 and 2 are swapped
// for the first product, which deals the hidden units to MFMA rows in the forward's bit-2/3-swapped order.  Image swizzles (16-byte
// slot of row r holds chunk slot ^ f(r)) are chosen so that the 32 lanes of a half read 32 distinct 8-byte bank pairs:
//   W2 chunk [256 d][64 hid], 128-byte rows:  f(d) = ((d >> 1) & 1) << 2         W1 chunk [64 hid][256 d], 512-byte rows:  f(u) = (u & 3) << 2
struct FfnBwdArgs {
    const bf16_t* ds16;
    const float* ds32;
    const bf16_t* w1;
    const bf16_t* w2;
    const uint32_t* bits;
    bf16_t* dhid;
    float* dx;
    int M, dff, Mp;
    // LNB: dx is the gradient wrt a LayerNorm output and nothing else reads it - the backward of that LayerNorm
    // (asr_add_layernorm_bwd's arithmetic) is the epilogue, on the rows as they leave the accumulators
    const float* ln_s;
    const float* ln_mean;
    const float* ln_rstd;
    const float* ln_gamma;
    const float* ln_beta;     // non-null: ln_s holds the LayerNorm's output y, x^ = (y - beta) / gamma
    const int32_t* row_len;
    int L;
    float* ds_out;
    bf16_t* ds16_out;
    float* dgamma;
    float* dbeta;
    float* dbias;
    asr_dropout_t drop_x;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x8 tr_pair(const unsigned char* p, int second_off) {
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + second_off));
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}

template <bool LNB>
__global__ __launch_bounds__(256, 1) void ffn_bwd_kernel(const FfnBwdArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[2 * W1BUF + 2 * W2BUF + HST_BYTES];
    unsigned char* const w2s = smem;                    // first product's weights here
    unsigned char* const w1s = smem + 2 * W2BUF;
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const hst = smem + 2 * W1BUF + 2 * W2BUF + wave * 4096;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * FBM + wave * 32 + r;
    const bool valid = m < a.M;
    const int mc = valid ? m : a.M - 1;
    const int dff = a.dff, NC = dff / FHC;

    const u32x4 rs1 = rsrc_words(a.w1, (unsigned)(dff * FD * 2)), rs2 = rsrc_words(a.w2, (unsigned)(dff * FD * 2));
    unsigned off1[8], off2[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = wave * 8 + k;
        {
            const int u = 2 * p + (lane >> 5), pc = lane & 31;
            off1[k] = (unsigned)(u * FD * 2 + ((pc ^ ((u & 3) << 2)) << 4));
        }
        {
            const int d = 8 * p + (lane >> 3), pc = lane & 7;
            off2[k] = (unsigned)(d * dff * 2 + ((pc ^ (((d >> 1) & 1) << 2)) << 4));
        }
    }
    auto dma_w1 = [&](int buf, int chunk, int j) {      // one 1-KiB piece of a chunk image (see the forward)
        dma16_asm(rs1, off1[j], chunk * (FHC * FD * 2), lds_addr_of(w1s + buf * W1BUF + (wave * 8 + j) * 1024));
    };
    auto dma_w2 = [&](int buf, int chunk, int j) {
        dma16_asm(rs2, off2[j], chunk * (FHC * 2), lds_addr_of(w2s + buf * W2BUF + (wave * 8 + j) * 1024));
    };

    // transposed-read addresses: within a 16-lane group, lane 4q + p supplies block row q, column quad p (first product: quad p')
    const int q = (lane & 15) >> 2, pq = lane & 3, g1 = (lane >> 4) & 1;
    const int pp = ((pq & 1) << 1) | (pq >> 1);
    unsigned a1[2], a2[4];
    {
        const unsigned l0 = (unsigned)((8 * h + q) * 128 + (2 * g1 + (pp >> 1)) * 16 + (pp & 1) * 8);
        a1[0] = l0 + (unsigned)((q >> 1) * 64);
        a1[1] = l0 + (unsigned)((1 - (q >> 1)) * 64);
#pragma unroll
        for (int v = 0; v < 4; ++v) a2[v] = (unsigned)((8 * h + q) * 512 + (4 * (v ^ q) + 2 * g1 + (pq >> 1)) * 16 + (pq & 1) * 8);
    }

    bf16x8 db[16];
    {
        const bf16_t* xr = a.ds16 + (int64_t)mc * FD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) db[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
#pragma unroll
    for (int j = 0; j < 8; ++j) dma_w2(0, 0, j);
    f32x16 Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[t][j] = 0.f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    u32x4 Hf[4];
    const auto rsh = __builtin_amdgcn_make_buffer_rsrc(a.dhid, 0, (int)((int64_t)a.M * dff * 2), 0x00020000);
    const u32x4 rsb = rsrc_words(a.bits, (unsigned)((int64_t)NC * 2 * a.Mp * 4));
    const unsigned boff = ((unsigned)h * a.Mp + mc) * 2u;      // uint16 [chunk][tile t][lane half h][token]: ffn2.hip's mask image
    const unsigned hwr = (unsigned)(r * 128), hrd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));      // as in the forward
    unsigned hoff[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int mt = blockIdx.x * FBM + wave * 32 + 8 * ps + (lane >> 3);
        hoff[ps] = mt < a.M ? (unsigned)mt * (unsigned)dff * 2u + 16u * (lane & 7) : 0x80000000u;
    }

    bf16x8 A[16];
    f32x4 res[32];
    const int m0 = blockIdx.x * FBM + wave * 32;
    auto frag1 = [&](const unsigned char* w2, int k) {      // first product, MFMA k: k-step k >> 1 (16 rows of the image), row tile k & 1
        return tr_pair(w2 + a1[k & 1] + (k >> 1) * 2048, 512);
    };
    auto frag2 = [&](const unsigned char* w1, int k) {      // second product, MFMA k: k-step k >> 3 (16 rows), row tile k & 7 of dX^T
        return tr_pair(w1 + a2[k & 3] + (k >> 3) * 8192 + ((k & 7) >> 2) * 256, 2048);
    };
    // element e = 16 t + j of a lane's chunk: the forward's uint16 of tile t (ffn2.hip) in half t of the word, bit j >> 1 (j even) / 8 + (j >> 1) (j odd)
    auto mask_pair = [&](const f32x16 (&S)[2], u32x4 (&Hn)[4], uint32_t word, int P) {
        const int t = P >> 3, p = P & 7;
        const float lo = drop_and(S[t][2 * p], word, 16 * t + p), hi = drop_and(S[t][2 * p + 1], word, 16 * t + 8 + p);
        uint32_t pk;
        asm("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(lo), "v"(hi));
        Hn[2 * t + (p >> 2)][p & 3] = pk;
    };
#define FFN_STEP() __builtin_amdgcn_sched_barrier(0)
#define FFN_WAIT_STAGE(NST)                                                           \
    do {                                                                              \
        asm volatile("s_waitcnt vmcnt(" #NST ") lgkmcnt(0)" ::: "memory");            \
        __builtin_amdgcn_s_barrier();                                                 \
        asm volatile("" ::: "memory");                                                \
    } while (0)

    auto body = [&](int i, u32x2 word_in, auto first_c, auto last_c) {
        uint32_t wlo = word_in[0], whi = word_in[1], word = 0;
        constexpr bool FIRST = decltype(first_c)::value, LAST = decltype(last_c)::value;
        const unsigned char* w2 = w2s + (i & 1) * W2BUF;
        const unsigned char* w1 = w1s + ((i - 1) & 1) * W1BUF;
        const int nxt = i + 1 < NC ? i + 1 : NC - 1;
        f32x16 S[2];
        u32x4 Hn[4], Hout[4];
        if constexpr (!LAST) {
#pragma unroll
            for (int t = 0; t < 2; ++t)
#pragma unroll
                for (int j = 0; j < 16; ++j) S[t][j] = 0.f;
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag1(w2, k);
            FFN_STEP();
#pragma unroll
            for (int k = 0; k < 32; ++k) {
                S[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], db[k >> 1], S[k & 1], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag1(w2, k + 8);
                else if (!FIRST) A[(k + 8) & 15] = frag2(w1, k + 8 - 32);
                if (k < 16) {      // this iteration's 16 LDS-DMA pieces, one per step: W2 of chunk i + 1, W1 of chunk i
                    if (k & 1) dma_w1(i & 1, i, k >> 1);
                    else dma_w2((i + 1) & 1, nxt, k >> 1);
                }
                if (!FIRST && k >= 24 && k < 28) Hout[k - 24] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 24) * 1024);
                FFN_STEP();
            }
            asm volatile("s_waitcnt vmcnt(16)" : "+v"(wlo), "+v"(whi));      // the two mask shorts: older than this iteration's 16 DMA pieces
            word = wlo | (whi << 16);
            FFN_STEP();
        } else {
#pragma unroll
            for (int k = 0; k < 8; ++k) A[k] = frag2(w1, k);
            FFN_STEP();
        }
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if constexpr (!FIRST) {
                Y[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 15], __builtin_bit_cast(bf16x8, Hf[k >> 3]), Y[k & 7], 0, 0, 0);
                if (k + 8 < 32) A[(k + 8) & 15] = frag2(w1, k + 8);
            }
            if constexpr (!LAST) {
                if ((k & 1) == 0) mask_pair(S, Hn, word, k >> 1);
                if (!FIRST && (k & 7) == 0) __builtin_amdgcn_raw_buffer_store_b128(Hout[k >> 3], rsh, hoff[k >> 3], (i - 1) * (FHC * 2), FFN_HID_POLICY);
                if ((k & 7) == 7) *reinterpret_cast<u32x4*>(hst + hwr + ((((k >> 3) * 2 + h) ^ (r & 7)) << 4)) = Hn[k >> 3];
            } else {
                if (k >= 4 && k < 8) Hout[k - 4] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 4) * 1024);
                if (k >= 16 && k < 20) __builtin_amdgcn_raw_buffer_store_b128(Hout[k - 16], rsh, hoff[k - 16], (i - 1) * (FHC * 2), FFN_HID_POLICY);
                // the epilogue's ds32 row k (row layout: lane = 4 columns of a token row), requested a half-iteration ahead of its use
                const int rowc = m0 + k < a.M ? m0 + k : a.M - 1;
                res[k] = *reinterpret_cast<const f32x4*>(a.ds32 + (int64_t)rowc * FD + 4 * lane);
            }
            FFN_STEP();
        }
        if constexpr (!LAST) {
#pragma unroll
            for (int sg = 0; sg < 4; ++sg) Hf[sg] = Hn[sg];
        }
    };
    {
        const u32x2 word = load16x2_asm(rsb, boff, 0, 4u * a.Mp);
        body(0, word, std::true_type{}, std::false_type{});
        FFN_WAIT_STAGE(0);
    }
    for (int i = 1; i < NC; ++i) {
        // this chunk's mask word: issued BEFORE the iteration's LDS-DMA, waited for (vmcnt(16)) where the second product begins
        const u32x2 word = load16x2_asm(rsb, boff, i * (2 * a.Mp * 4), i * (2 * a.Mp * 4) + 4u * a.Mp);
        body(i, word, std::false_type{}, std::false_type{});
        FFN_WAIT_STAGE(4);
    }
    body(NC, u32x2{0u, 0u}, std::false_type{}, std::true_type{});
#undef FFN_WAIT_STAGE
#undef FFN_STEP

    // ---- epilogue: dx = dX + ds32, through the wave's 32-KiB LDS tile (see the forward) so that loads and stores are whole rows --------
    __syncthreads();
    unsigned char* const tile = smem + wave * 32768;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(tile + r * 1024 + (((8 * t + 2 * g + h) ^ (r & 7)) << 4)) =
                f32x4{Y[t][4 * g], Y[t][4 * g + 1], Y[t][4 * g + 2], Y[t][4 * g + 3]};
    if constexpr (!LNB) {
        const auto rsx = __builtin_amdgcn_make_buffer_rsrc(a.dx, 0, (int)((int64_t)a.M * FD * 4), 0x00020000);
#pragma unroll
        for (int tr = 0; tr < 32; ++tr) {
            const f32x4 y = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (tr & 7)) << 4));
            const f32x4 v = y + res[tr];
            const int row = m0 + tr;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rsx, row < a.M ? (unsigned)row * (FD * 4u) + 16u * lane : 0x80000000u, 0, 0);
        }
    } else {
        // ---- ... and the backward of the LayerNorm whose output this sub-layer read (attention.py:60 in the encoder layer): a row lies
        // across the wave exactly as add_layernorm_bwd_kernel holds it (lane = 4 columns), so this is that kernel's arithmetic on rows
        // that never went to memory as dx:  d = dx (0 for masked rows);  xh = (s - mean) * rstd;  g = d * gamma;
        //   ds = (g - mean(g) - xh * mean(g * xh)) * rstd;   dgamma += d * xh, dbeta += d, dbias += dropout_x(ds)   (column sums)
        // The 32 rows of the pre-norm sum are requested up front (the accumulators' registers are free now).
        f32x4 srow[32];
#pragma unroll
        for (int tr = 0; tr < 32; ++tr) {
            const int rowc = m0 + tr < a.M ? m0 + tr : a.M - 1;
            srow[tr] = *reinterpret_cast<const f32x4*>(a.ln_s + (int64_t)rowc * FD + 4 * lane);
        }
        const int myrow = m0 + r < a.M ? m0 + r : a.M - 1;                    // lane (and lane + 32) keep row r's statistics
        const float mu_l = a.ln_mean ? a.ln_mean[myrow] : 0.f, rs_l = a.ln_rstd[myrow];
        const int b_l = myrow / a.L, t_l = myrow - b_l * a.L;
        const int keep_l = (m0 + r < a.M && t_l < (a.row_len ? a.row_len[b_l] : a.L)) ? 1 : 0;
        const f32x4 gam = *reinterpret_cast<const f32x4*>(a.ln_gamma + 4 * lane);
        const f32x4 betav = a.ln_beta ? *reinterpret_cast<const f32x4*>(a.ln_beta + 4 * lane) : f32x4{0, 0, 0, 0};
        const asr_dropout_t drop = drop_resolve(a.drop_x);
        const float scx = drop_scale(drop);
        const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.ds_out, 0, (int)((int64_t)a.M * FD * 4), 0x00020000);
        const auto rsh16 = __builtin_amdgcn_make_buffer_rsrc(a.ds16_out, 0, (int)((int64_t)a.M * FD * 2), 0x00020000);
        f32x4 ag = {0, 0, 0, 0}, ab = {0, 0, 0, 0}, as = {0, 0, 0, 0};
        int bb = m0 / a.L, tt = m0 - bb * a.L;                                 // (utterance, frame) of row m0 + tr, kept in step
        constexpr int RW = 4;                                                 // rows in flight together: their reductions interleave
#pragma unroll
        for (int tp = 0; tp < 32; tp += RW) {
            f32x4 d[RW], xh[RW], g[RW];
            float s1[RW], s2[RW], rs[RW];
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int tr = tp + u;
                const f32x4 y = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (tr & 7)) << 4));
                const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu_l), tr));
                rs[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs_l), tr));
                const bool keep = __builtin_amdgcn_readlane(keep_l, tr) != 0;
                d[u] = keep ? y + res[tr] : f32x4{0, 0, 0, 0};
                if (a.ln_beta) {      // ln_s is the LayerNorm's OUTPUT (the forward kept no pre-norm sum): x^ = (y - beta) / gamma
#pragma unroll
                    for (int e = 0; e < 4; ++e) xh[u][e] = (keep && gam[e] != 0.f) ? (srow[tr][e] - betav[e]) / gam[e] : 0.f;
                } else
                    xh[u] = (srow[tr] - mu) * rs[u];
                ag += d[u] * xh[u];
                ab += d[u];
                g[u] = d[u] * gam;
                const f32x4 gx = g[u] * xh[u];
                s1[u] = (g[u][0] + g[u][1]) + (g[u][2] + g[u][3]);
                s2[u] = (gx[0] + gx[1]) + (gx[2] + gx[3]);
            }
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                s1[u] = wave_sum_dpp(s1[u]);
                s2[u] = wave_sum_dpp(s2[u]);
            }
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int row = m0 + tp + u;
                const float m1 = s1[u] * (1.f / FD), m2 = s2[u] * (1.f / FD);
                f32x4 o = (g[u] - m1 - xh[u] * m2) * rs[u];
                const unsigned off = row < a.M ? (unsigned)row * FD : 0x20000000u;      // (in elements; past the buffer for a row that is not there)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, off * 4u + 16u * lane, 0, 0);     // gradient wrt the residual
                if (drop.thr16)                                                         // gradient wrt the projection's output (dropout's input)
                    o = drop4(drop, drop_subkey(drop, (uint32_t)bb), (uint32_t)tt, FD >> 1, 4 * lane, o, scx);
                as += o;
                const bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rsh16, off * 2u + 8u * lane, 0, 0);
                if (++tt == a.L) { tt = 0; ++bb; }
            }
        }
        // column sums: the four waves' partials meet in LDS (the tiles are done with), one atomic per column and workgroup
        __syncthreads();
        float* red = reinterpret_cast<float*>(smem);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            red[(0 * 4 + wave) * FD + 4 * lane + k] = ag[k];
            red[(1 * 4 + wave) * FD + 4 * lane + k] = ab[k];
            red[(2 * 4 + wave) * FD + 4 * lane + k] = as[k];
        }
        __syncthreads();
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            float* dst = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.dbias);
            if (!dst) continue;
            const float v = (red[(which * 4 + 0) * FD + tid] + red[(which * 4 + 1) * FD + tid]) + (red[(which * 4 + 2) * FD + tid] + red[(which * 4 + 3) * FD + tid]);
            atomicAdd(dst + tid, v);
        }
    }
}

// ---- head-major projections on the same structure (attention.py:43-49: w_qs / w_ks / w_vs + view / permute / contiguous) ---------
// out[p][b][head][l][0..64) = (x[b, l, :] . W[(p * h + head) * 64 + j, :] + bias) [* scale for p == 0]: a 64-unit chunk of the
// feed-forward kernel's first product is exactly one head of one projection, and the lane-owned 16-byte pieces of the chunk leave
// through the same per-wave LDS tile as the hidden activation does there - 128-byte token rows, consecutive tokens of an utterance
// consecutive in memory.  x is read once for all n_proj * h chunks (the tiled GEMM re-reads it per column block).  At one wave per
// SIMD nothing hides a wait, so nothing in a chunk may wait: the weights stream through a ring of FOUR 32-KiB images requested three
// chunks ahead (the fragment ring runs across chunk borders, and a chunk's stores have two chunk times to be acknowledged before a
// wait counts them), the bias joins in the pack (one fma per value on the pre-scaled staged copy) and the previous chunk's
// accumulators leave beside this chunk's MFMAs.  Measured at M = 32000: 22-23 us for Q/K/V (N = 768; tiled GEMM 24.8) and 72 us for
// the decoder's 6-layer cross K/V (N = 3072; tiled 88-92) - the loop rebuilt with pieces left out (tools/ablate_heads.sh): MFMAs
// alone 0.45 us per chunk (the matrix pipe's own time), + fragment ring and LDS-DMA 0.68, + pack / tile / stores 1.46: the parts
// still add up.  (Tried: 8 waves, wave w and w + 4 on one SIMD taking turns - one multiplies chunk p while the other packs and
// stores chunk p - 1, a workgroup barrier per phase: correct, but 27 / 81 us: a multiply phase alone takes 1.07 us and a pack phase
// alone 0.95 us - per-phase latencies (exposed first fragments, LDS round trips of the tile, store issue, barrier) that overlap
// only with the OTHER phase, so the matrix pipe is busy a third of a phase.)
struct HeadsArgs {
    const bf16_t* x16;
    const bf16_t* w;
    const float* bias;
    bf16_t* out;
    int M, L, h, nch, nscaled, N;
    long long proj_stride;      // elements between projections
    float scale;
};
constexpr int HEADS_MAX_N = 3072;
constexpr int HEADS_SMEM = 4 * W1BUF + HST_BYTES + HEADS_MAX_N * 4;

// ABL (timing diagnostics only, wrong results): 1 = no LDS-DMA in the loop, 2 = no fragment reads, 4 = nothing leaves, 8 = no MFMA
template <int ABL = 0>
__global__ __launch_bounds__(256, 1) void proj_heads_rows_kernel(const HeadsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[HEADS_SMEM];
    unsigned char* const w1s = smem;
    float* const b1s = reinterpret_cast<float*>(smem + 4 * W1BUF + HST_BYTES);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    unsigned char* const hst = smem + 4 * W1BUF + wave * 4096;
    const int r = lane & 31, h = lane >> 5;
    const int m = blockIdx.x * FBM + wave * 32 + r;
    const int mc = m < a.M ? m : a.M - 1;
    const int NC = a.nch;

    const auto rs1 = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a.w), 0, a.N * FD * 2, 0x00020000);
    unsigned off1[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) {       // (the feed-forward kernel's W1 image: piece p = rows 2p, 2p + 1 of the chunk; 16-byte slot pc of row u holds chunk (pc & 16) | ((pc ^ u) & 15))
        const int p = wave * 8 + k, u = 2 * p + (lane >> 5), pc = lane & 31;
        off1[k] = (unsigned)(u * FD * 2 + ((pc & 16) | ((pc ^ u) & 15)) * 16);
    }
    auto dma_w = [&](int buf, int chunk, int j) {
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs1, (lds_void*)(w1s + buf * W1BUF + (wave * 8 + j) * 1024), 16, off1[j], chunk * (FHC * FD * 2), 0, 0);
    };
    const int ur = swap23(r), u15 = ur & 15;      // MFMA row r holds unit swap23(r) of the chunk: a lane's 16 accumulator registers are two runs of 8 consecutive units
    unsigned a1[8];
#pragma unroll
    for (int kk = 0; kk < 8; ++kk) a1[kk] = (unsigned)(ur * 512 + (((2 * kk + h) ^ u15) << 4));

    bf16x8 xb[16];
    {
        const bf16_t* xr = a.x16 + (int64_t)mc * FD + 8 * h;
#pragma unroll
        for (int ks = 0; ks < 16; ++ks) xb[ks] = *reinterpret_cast<const bf16x8*>(xr + 16 * ks);
    }
    for (int i = tid * 4; i < a.N; i += 1024) {      // (pre-scaled where the output is: the pack is one fma per register pair)
        const f32x4 bv = a.bias ? *reinterpret_cast<const f32x4*>(a.bias + i) : f32x4{0.f, 0.f, 0.f, 0.f};
        *reinterpret_cast<f32x4*>(b1s + i) = i < a.nscaled * FHC ? bv * a.scale : bv;
    }
#pragma unroll
    for (int c = 0; c < 3; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) dma_w(c, c < NC ? c : NC - 1, j);

    // output addressing: token mt of this wave's store slot ps -> (b, l); one 128-byte row per (token, chunk)
    const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out, 0, (int)((int64_t)(NC / a.h) * a.proj_stride * 2), 0x00020000);
    const unsigned hwr = (unsigned)(r * 128), hrd = (unsigned)((lane >> 3) * 128 + (((lane & 7) ^ ((lane >> 3) & 7)) << 4));
    unsigned hoff[4];
#pragma unroll
    for (int ps = 0; ps < 4; ++ps) {
        const int mt = blockIdx.x * FBM + wave * 32 + 8 * ps + (lane >> 3);
        const int b = mt / a.L, l = mt - b * a.L;
        hoff[ps] = mt < a.M ? ((unsigned)(b * a.h) * (unsigned)a.L + (unsigned)l) * 128u + 16u * (lane & 7) : 0x80000000u;
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();

    auto frag1 = [&](const unsigned char* w1, int k) {
        const int ks = k >> 1, t = k & 1;
        return *reinterpret_cast<const bf16x8*>(w1 + a1[ks & 7] + t * 16384 + (ks >> 3) * 256);
    };
    // piece i (0..7) of a chunk's bias in accumulator layout: tile i >> 2, registers 4 (i & 3) .. + 4 = register pairs 2 i, 2 i + 1
    auto bias_piece = [&](int chunk, int i) {
        return *reinterpret_cast<const f32x4*>(b1s + chunk * FHC + 8 * h + 32 * (i >> 2) + ((i & 3) >> 1) * 16 + (i & 1) * 4);
    };
    auto out_soff = [&](int chunk) {       // byte offset of (projection, head) = chunk
        const int p = chunk / a.h, hd = chunk - p * a.h;
        return (unsigned)((int64_t)p * a.proj_stride * 2 + (int64_t)hd * a.L * 128);
    };
#define HEADS_STEP() __builtin_amdgcn_sched_barrier(0)
    typedef __attribute__((ext_vector_type(2))) float f32x2_t;
    bf16x8 A[8];        // fragment ring: MFMA k takes A[k & 7], which is requested again for MFMA k + 8 right behind it
    u32x4 Hn[4], Hout[4];
    f32x4 Bq[4];        // bias pieces on their way to the pack
    // register pair P of a finished chunk: (acc * scale + bias * scale) -> two bf16 (the staged bias is pre-scaled).  Three stages, one
    // step apart, so that no instruction of a step waits for another of the same step (at one wave per SIMD a dependent VALU group
    // costs its latency: tools/probe/clock_probe.hip modes 22 / 25 - 34 against 44 cycles per one-MFMA step):
    //   fetch: the pair into plain VGPRs (the accumulators live in AGPRs)   fma: * scale + bias   pack: cvt_pk into the tile word
    float st0[16], st1[16];
    f32x2_t fv[16];
    auto pack_fetch = [&](const f32x16 (&S)[2], int P) {
        st0[P] = S[P >> 3][2 * (P & 7)];
        st1[P] = S[P >> 3][2 * (P & 7) + 1];
        asm volatile("" : "+v"(st0[P]), "+v"(st1[P]));
    };
    auto pack_fma = [&](int P, f32x2_t sc2) {
        const f32x4 bq = Bq[(P >> 1) & 3];
        fv[P] = __builtin_elementwise_fma(f32x2_t{st0[P], st1[P]}, sc2, f32x2_t{bq[(P & 1) * 2], bq[(P & 1) * 2 + 1]});
    };
    auto pack_cvt = [&](int P) {
        uint32_t pk;
        asm volatile("v_cvt_pk_bf16_f32 %0, %1, %2" : "=v"(pk) : "v"(fv[P][0]), "v"(fv[P][1]));
        Hn[P >> 2][P & 3] = pk;
    };
    const f32x16 zero16 = {0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f, 0.f};
    // chunk c: 32 MFMAs into Sc; beside them the previous chunk's accumulators Sp leave (pack: steps 0..15 with the bias pieces read
    // two steps ahead, tile: 3 / 7 / 11 / 15, row-wise read-back: 20..23, stores: 24 / 26 / 28 / 30), and from step 24 the fragments
    // requested are the next chunk's.  VMEM order: 8 LDS-DMA pieces (chunk c + 3), then the 4 stores.
    // Register budget: everything here has to fit the 256 architectural VGPRs - what does not is parked in AGPRs and every
    // v_accvgpr move runs in series with the MFMAs (measured: 160 of them per chunk cost more than the chunk's 32 MFMAs).
    auto body = [&](int c, f32x16 (&Sc)[2], f32x16 (&Sp)[2], auto first_c) {
        constexpr bool FIRST = decltype(first_c)::value;
        const unsigned char* wc = w1s + (c & 3) * W1BUF;
        const unsigned char* wn = w1s + ((c + 1) & 3) * W1BUF;
        const int bd = (c + 3) & 3, cd = c + 3 < NC ? c + 3 : NC - 1;
        const unsigned so = FIRST ? 0u : out_soff(c - 1);
        const float sc = (!FIRST && c - 1 < a.nscaled) ? a.scale : 1.0f;
        const f32x2_t sc2 = {sc, sc};
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            if (!(ABL & 8)) Sc[k & 1] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(A[k & 7], xb[k >> 1], k < 2 ? zero16 : Sc[k & 1], 0, 0, 0);
            if (!(ABL & 2)) A[k & 7] = k + 8 < 32 ? frag1(wc, k + 8) : frag1(wn, k + 8 - 32);
            if (k < 16 && !(k & 1) && !(ABL & 1)) dma_w(bd, cd, k >> 1);
            if constexpr (!FIRST && !(ABL & 4)) {
                if (k < 12 && !(k & 1)) Bq[((k >> 1) + 2) & 3] = bias_piece(c - 1, (k >> 1) + 2);
                if (k == 0) pack_fetch(Sp, 0);
                if (k >= 1 && k < 17) pack_cvt(k - 1);
                if (k < 16) pack_fma(k, sc2);
                if (k + 1 < 16) pack_fetch(Sp, k + 1);
                if (k >= 4 && k < 17 && (k & 3) == 0 && !(ABL & 32)) *reinterpret_cast<u32x4*>(hst + hwr + ((((((k >> 2) - 1) * 2 + h)) ^ (r & 7)) << 4)) = Hn[(k >> 2) - 1];
                if (k >= 20 && k < 24 && !(ABL & 32)) Hout[k - 20] = *reinterpret_cast<const u32x4*>(hst + hrd + (k - 20) * 1024);
                if (k >= 24 && !(k & 1) && !(ABL & 16)) __builtin_amdgcn_raw_buffer_store_b128((ABL & 32) ? Hn[(k - 24) >> 1] : Hout[(k - 24) >> 1], rso, hoff[(k - 24) >> 1], so, 0);
                if ((ABL & 16) && k == 31) asm volatile("" :: "v"(Hout[0]), "v"(Hout[1]), "v"(Hout[2]), "v"(Hout[3]), "v"(Hn[0]), "v"(Hn[1]), "v"(Hn[2]), "v"(Hn[3]));
            }
            if (k == 28 || k == 30) Bq[(k - 28) >> 1] = bias_piece(c, (k - 28) >> 1);      // the first two pieces of THIS chunk, for the next body
            HEADS_STEP();
        }
        // chunk c + 2's image (requested one chunk ago, read from step 24 of the next chunk) has to be there; everything younger may
        // stay in flight: this chunk's 8 pieces and 4 stores, the previous chunk's 4 stores.  (Waiting for the stores themselves ran
        // the launch at their latency.)
        if constexpr (!FIRST && !(ABL & 21)) {
            if (c == 1) asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(16)" ::: "memory");
        } else if constexpr (ABL != 0) {
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
    };
    f32x16 SA[2], SB[2];
#pragma unroll
    for (int k = 0; k < 8; ++k) A[k] = frag1(w1s, k);
    HEADS_STEP();
    body(0, SA, SB, std::true_type{});
    int c = 1;
    for (; c + 1 < NC; c += 2) {
        body(c, SB, SA, std::false_type{});
        body(c + 1, SA, SB, std::false_type{});
    }
#undef HEADS_STEP
    // the last chunk's accumulators
    auto drain = [&](f32x16 (&Sp)[2]) {
        const unsigned so = out_soff(NC - 1);
        const float sc = NC - 1 < a.nscaled ? a.scale : 1.0f;
        const f32x2_t sc2 = {sc, sc};
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            if (!(k & 1)) Bq[(k >> 1) & 3] = bias_piece(NC - 1, k >> 1);
            pack_fetch(Sp, k);
            pack_fma(k, sc2);
            pack_cvt(k);
            if ((k & 3) == 3) *reinterpret_cast<u32x4*>(hst + hwr + ((((k >> 2) * 2 + h) ^ (r & 7)) << 4)) = Hn[k >> 2];
        }
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) Hout[ps] = *reinterpret_cast<const u32x4*>(hst + hrd + ps * 1024);
#pragma unroll
        for (int ps = 0; ps < 4; ++ps) __builtin_amdgcn_raw_buffer_store_b128(Hout[ps], rso, hoff[ps], so, 0);
    };
    if (c < NC) {
        body(c, SB, SA, std::false_type{});
        drain(SB);
    } else {
        drain(SA);
    }
}

}  // namespace

// asr_proj_heads's encoder-sized bf16 case (gemm.hip dispatches here): 0 = launched, -2 = not this kernel's shape
int asr_proj_heads_rows(hipStream_t stream, const void* X, const void* W, const float* bias, void* out, int64_t proj_stride, int n_proj,
                        int B, int L, int h, float scale_first) {
    const int64_t M64 = (int64_t)B * L, N64 = (int64_t)n_proj * h * 64;
    if (N64 > HEADS_MAX_N || M64 * h * 128 >= (1ll << 31) || (int64_t)n_proj * proj_stride * 2 >= (1ll << 31)) return -2;
    HeadsArgs a{(const bf16_t*)X, (const bf16_t*)W, bias, (bf16_t*)out, (int)M64, L, h, n_proj * h, scale_first != 1.0f ? h : 0, (int)N64,
                (long long)proj_stride, scale_first};
    const dim3 grid((unsigned)((M64 + FBM - 1) / FBM));
    hipLaunchKernelGGL(proj_heads_rows_kernel<0>, grid, dim3(256), 0, stream, a);
    ASR_LAUNCH_CHECK("asr_proj_heads(rows)");
    return 0;
}


extern "C" int64_t asr_ffn_bits_words(int M, int d_ff) { return (int64_t)(d_ff / FHC) * 2 * ((M + FBM - 1) / FBM * FBM); }

extern "C" int asr_ffn_fwd(void* stream, const void* x16, const float* x32, const void* w1, const float* b1, const void* w2, const float* b2,
                           const float* gamma, const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out,
                           float* y32, void* y16, float* mean_out, float* rstd_out, int B, int L, int d_model, int d_ff, float eps,
                           asr_dropout_t drop_x) {
    const int64_t M64 = (int64_t)B * L;
    ASR_REQUIRE(d_model == FD, -1, "asr_ffn_fwd: d_model = %d (the fused sub-layer is built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_ffn_fwd: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(M64 > 0 && M64 * d_ff * 2 < (1ll << 31), -1, "asr_ffn_fwd: B * L out of range");
    ASR_REQUIRE(x16 && x32 && w1 && b1 && w2 && b2 && gamma && beta && y32, -1, "asr_ffn_fwd: null argument");
    ASR_REQUIRE((hid_out == nullptr) == (bits_out == nullptr), -1, "asr_ffn_fwd: hid_out and bits_out come together (training) or not at all");
    ASR_REQUIRE(hid_out || (!s_out && !mean_out && !rstd_out), -1, "asr_ffn_fwd: s_out / mean_out / rstd_out are training outputs (pass hid_out and bits_out too)");
    ASR_REQUIRE(asr_aligned(x16, 16) && asr_aligned(x32, 16) && asr_aligned(w1, 16) && asr_aligned(w2, 16) && asr_aligned(y32, 16) &&
                    asr_aligned(hid_out, 16) && asr_aligned(s_out, 16) && asr_aligned(y16, 8) && asr_aligned(b1, 16) && asr_aligned(b2, 16) &&
                    asr_aligned(gamma, 16) && asr_aligned(beta, 16), -1, "asr_ffn_fwd: 16-byte aligned buffers required");
    ASR_REQUIRE(asr_aligned(bits_out, 4), -1, "asr_ffn_fwd: bits_out must be 4-byte aligned");
    // the two-waves-per-SIMD form (ffn2.hip: generated loop + this sub-layer's epilogue)
    return asr_ffn_fwd2_launch((hipStream_t)stream, x16, x32, w1, b1, w2, b2, gamma, beta, row_len, hid_out, bits_out, s_out, y32, y16, mean_out,
                               rstd_out, (int)M64, L, d_ff, eps, drop_x);
}

extern "C" int asr_attn_ffn_fwd(void* stream, const void* ctx16, const float* residual, const void* wo, const float* bo, const float* gamma0,
                                const float* beta0, float eps0, asr_dropout_t drop0, float* s0_out, float* x32, void* x16, float* mean0_out,
                                float* rstd0_out, const void* w1, const float* b1, const void* w2, const float* b2, const float* gamma,
                                const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out, float* y32, void* y16,
                                float* mean_out, float* rstd_out, int B, int L, int d_model, int d_ff, float eps, asr_dropout_t drop_x) {
    const int64_t M64 = (int64_t)B * L;
    ASR_REQUIRE(d_model == FD, -1, "asr_attn_ffn_fwd: d_model = %d (the fused sub-layers are built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_attn_ffn_fwd: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(M64 > 0 && M64 * d_ff * 2 < (1ll << 31), -1, "asr_attn_ffn_fwd: B * L out of range");
    ASR_REQUIRE(ctx16 && residual && wo && bo && gamma0 && beta0 && x32 && x16 && w1 && b1 && w2 && b2 && gamma && beta && y32, -1,
                "asr_attn_ffn_fwd: null argument");
    ASR_REQUIRE((hid_out == nullptr) == (bits_out == nullptr), -1, "asr_attn_ffn_fwd: hid_out and bits_out come together (training) or not at all");
    ASR_REQUIRE(hid_out || (!s_out && !mean_out && !rstd_out && !s0_out && !mean0_out && !rstd0_out), -1,
                "asr_attn_ffn_fwd: s / mean / rstd are training outputs (pass hid_out and bits_out too)");
    ASR_REQUIRE((drop0.thr16 != 0) == (drop_x.thr16 != 0), ASR_ERR_UNSUPPORTED, "asr_attn_ffn_fwd: both dropout sites active or neither");
    ASR_REQUIRE(drop0.thr16 < 65536u && drop_x.thr16 < 65536u, ASR_ERR_ARG, "asr_attn_ffn_fwd: dropout thr16 must be < 65536");
    ASR_REQUIRE(asr_aligned(ctx16, 16) && asr_aligned(residual, 16) && asr_aligned(wo, 16) && asr_aligned(bo, 16) && asr_aligned(gamma0, 16) &&
                    asr_aligned(beta0, 16) && asr_aligned(s0_out, 16) && asr_aligned(x16, 16) && asr_aligned(x32, 16) && asr_aligned(w1, 16) &&
                    asr_aligned(w2, 16) && asr_aligned(y32, 16) && asr_aligned(hid_out, 16) && asr_aligned(s_out, 16) && asr_aligned(y16, 8) &&
                    asr_aligned(b1, 16) && asr_aligned(b2, 16) && asr_aligned(gamma, 16) && asr_aligned(beta, 16),
                -1, "asr_attn_ffn_fwd: 16-byte aligned buffers required");
    ASR_REQUIRE(asr_aligned(bits_out, 4), -1, "asr_attn_ffn_fwd: bits_out must be 4-byte aligned");
    const asr_ffn2_pre_t pre{ctx16, residual, wo, bo, gamma0, beta0, s0_out, mean0_out, rstd0_out, eps0, drop0};
    return asr_ffn_fwd2_launch((hipStream_t)stream, x16, x32, w1, b1, w2, b2, gamma, beta, row_len, hid_out, bits_out, s_out, y32, y16, mean_out,
                               rstd_out, (int)M64, L, d_ff, eps, drop_x, &pre);
}

extern "C" int asr_proj_ln_fwd(void* stream, const void* ctx16, const float* residual, const void* w, const float* bias, const float* gamma,
                               const float* beta, const int32_t* row_len, float* s_out, float* y32, void* y16, float* mean_out,
                               float* rstd_out, int B, int L, int d_model, float eps, asr_dropout_t drop_x) {
    const int64_t M64 = (int64_t)B * L;
    ASR_REQUIRE(d_model == FD, -1, "asr_proj_ln_fwd: d_model = %d (built for 256 = h * d_v inputs and 256 outputs)", d_model);
    ASR_REQUIRE(M64 > 0 && M64 * FD * 4 < (1ll << 31), -1, "asr_proj_ln_fwd: B * L out of range");
    ASR_REQUIRE(ctx16 && residual && w && bias && gamma && beta && y32, -1, "asr_proj_ln_fwd: null argument");
    const bool train = s_out || mean_out || rstd_out;      // (s_out alone may be NULL in training: the backward then takes x^ from y32)
    ASR_REQUIRE(asr_aligned(ctx16, 16) && asr_aligned(residual, 16) && asr_aligned(w, 16) && asr_aligned(y32, 16) && asr_aligned(s_out, 16) &&
                    asr_aligned(y16, 8) && asr_aligned(bias, 16) && asr_aligned(gamma, 16) && asr_aligned(beta, 16), -1,
                "asr_proj_ln_fwd: 16-byte aligned buffers required");
    const int M = (int)M64;
    ProjLnArgs a{(const bf16_t*)ctx16, residual, (const bf16_t*)w, bias, gamma, beta, row_len, s_out, y32, (bf16_t*)y16, mean_out, rstd_out, M, L,
                 eps, drop_x};
    const dim3 grid((M + FBM - 1) / FBM), block(256);
    const bool dr = drop_x.thr16 != 0;
    if (train && dr) hipLaunchKernelGGL((proj_ln_kernel<true, true>), grid, block, 0, (hipStream_t)stream, a);
    else if (train) hipLaunchKernelGGL((proj_ln_kernel<true, false>), grid, block, 0, (hipStream_t)stream, a);
    else if (dr) hipLaunchKernelGGL((proj_ln_kernel<false, true>), grid, block, 0, (hipStream_t)stream, a);
    else hipLaunchKernelGGL((proj_ln_kernel<false, false>), grid, block, 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_proj_ln_fwd");
    return 0;
}

extern "C" int asr_ffn_bwd(void* stream, const void* ds16, const float* ds32, const void* w1, const void* w2, const void* bits,
                           void* dhid_out, float* dx_out, int M, int d_model, int d_ff) {
    ASR_REQUIRE(d_model == FD, -1, "asr_ffn_bwd: d_model = %d (the fused sub-layer is built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_ffn_bwd: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(M > 0 && (int64_t)M * d_ff * 2 < (1ll << 31), -1, "asr_ffn_bwd: M out of range");
    ASR_REQUIRE(ds16 && ds32 && w1 && w2 && bits && dhid_out && dx_out, -1, "asr_ffn_bwd: null argument");
    ASR_REQUIRE(asr_aligned(ds16, 16) && asr_aligned(ds32, 16) && asr_aligned(w1, 16) && asr_aligned(w2, 16) && asr_aligned(dhid_out, 16) &&
                    asr_aligned(dx_out, 16), -1, "asr_ffn_bwd: 16-byte aligned buffers required");
    FfnBwdArgs a{};
    a.ds16 = (const bf16_t*)ds16; a.ds32 = ds32; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.bits = (const uint32_t*)bits;
    a.dhid = (bf16_t*)dhid_out; a.dx = dx_out; a.M = M; a.dff = d_ff; a.Mp = (M + FBM - 1) / FBM * FBM;
    hipLaunchKernelGGL(ffn_bwd_kernel<false>, dim3((M + FBM - 1) / FBM), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_ffn_bwd");
    return 0;
}

extern "C" int asr_ffn_bwd_ln(void* stream, const void* ds16, const float* ds32, const void* w1, const void* w2, const void* bits,
                              void* dhid_out, int B, int L, int d_model, int d_ff, const float* ln_s, const float* ln_mean,
                              const float* ln_rstd, const float* ln_gamma, const float* ln_beta, const int32_t* row_len, float* ds_out,
                              void* ds16_out, float* dgamma, float* dbeta, float* dbias, asr_dropout_t drop_x) {
    ASR_REQUIRE(d_model == FD, -1, "asr_ffn_bwd_ln: d_model = %d (the fused sub-layer is built for 256)", d_model);
    ASR_REQUIRE(d_ff >= FHC && d_ff % FHC == 0 && d_ff <= FFN_MAX_DFF, -1, "asr_ffn_bwd_ln: d_ff = %d (a multiple of 64 up to %d)", d_ff, FFN_MAX_DFF);
    ASR_REQUIRE(B > 0 && L > 0 && (int64_t)B * L * d_ff * 2 < (1ll << 31), -1, "asr_ffn_bwd_ln: B * L out of range");
    const int M = B * L;
    ASR_REQUIRE(ds16 && ds32 && w1 && w2 && bits && dhid_out && ln_s && (ln_mean || ln_beta) && ln_rstd && ln_gamma && ds_out && ds16_out && dgamma && dbeta,
                -1, "asr_ffn_bwd_ln: null argument");
    ASR_REQUIRE(asr_aligned(ds16, 16) && asr_aligned(ds32, 16) && asr_aligned(w1, 16) && asr_aligned(w2, 16) && asr_aligned(dhid_out, 16) &&
                    asr_aligned(ln_s, 16) && asr_aligned(ln_gamma, 16) && asr_aligned(ds_out, 16) && asr_aligned(ds16_out, 16),
                -1, "asr_ffn_bwd_ln: 16-byte aligned buffers required");
    FfnBwdArgs a{};
    a.ds16 = (const bf16_t*)ds16; a.ds32 = ds32; a.w1 = (const bf16_t*)w1; a.w2 = (const bf16_t*)w2; a.bits = (const uint32_t*)bits;
    a.dhid = (bf16_t*)dhid_out; a.dx = nullptr; a.M = M; a.dff = d_ff; a.Mp = (M + FBM - 1) / FBM * FBM;
    a.ln_s = ln_s; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.row_len = row_len; a.L = L;
    a.ds_out = ds_out; a.ds16_out = (bf16_t*)ds16_out; a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.drop_x = drop_x;
    hipLaunchKernelGGL(ffn_bwd_kernel<true>, dim3((M + FBM - 1) / FBM), dim3(256), 0, (hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_ffn_bwd_ln");
    return 0;
}
