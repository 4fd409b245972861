// Position-wise feed-forward sub-layer of an encoder layer at TWO waves per SIMD (d_model = 256): the forward launch.
//   y = LayerNorm(dropout(relu(x W1^T + b1) W2^T + b2) + x) [* non_pad_mask]                    src/transformer/module.py:48-53, encoder.py:77
// Same decomposition as ffn.hip (a workgroup owns 128 tokens and streams W1 / W2 in 64-unit chunks through LDS-DMA double buffers,
// the hidden activation never makes a round trip through HBM), re-cut for eight waves: waves wv and wv + 4 share a SIMD and the 32
// tokens of pair wv & 3; per chunk each takes a 32-unit half of the first product and a 128-row half of the second, the ReLU-ed halves
// cross through a 4-KiB LDS tile per pair.  The loop is a generated instruction stream (tools/gen_ffn_fwd.py -> ffn_fwd2_asm.inc:
// fixed registers, every LDS-DMA request / fragment read / ReLU slice placed in an MFMA gap); this file is its C++ frame:
//   prologue  every lane-dependent address of the loop, computed here with ffn.hip's formulas and left in LDS (one dword per thread
//             and parameter) - the asm block has no VGPR inputs, so the compiler's registers and the block's 176 + 64 never compete
//   epilogue  bias + dropout + residual + LayerNorm + row mask over whole token rows read back from the pair's LDS tile
//             (add_layernorm_fwd_kernel's arithmetic; outputs as asr_gemm_nt x 2 + asr_add_layernorm_fwd leave them)
// Mask image (training, private between this launch and asr_ffn_bwd*): uint16 [chunk][half w][lane half h][token], bit p = unit 2 p of
// the lane's 16 units of the half positive, bit 8 + p = unit 2 p + 1 (a lane's units: e = register index of the 32 x 32 accumulator).
#include "asr_common.h"

#include <stddef.h>
#include <stdlib.h>

#include "ffn_fwd2_params.h"
#ifndef FFN2_FWD_INC
#define FFN2_FWD_INC "ffn_fwd2_asm.inc"
#endif
#include FFN2_FWD_INC

namespace {

constexpr int FBM = 128, FHC = 64, FD = 256;
constexpr int W2RING = 65536, HXRING = 131072;
constexpr int SMEM2 = 163840;       // W1 ring 2 x 32 KiB | W2 ring 2 x 32 KiB (first: the parameter area) | H tiles [2][4][4 KiB]

__device__ __forceinline__ int swap23(int r) { return (r & ~12) | ((r & 4) << 1) | ((r & 8) >> 1); }

template <int CTRL> __device__ __forceinline__ float dpp_perm(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_dpp(float v) {
    v += dpp_perm<0xB1>(v);
    v += dpp_perm<0x4E>(v);
    v += dpp_perm<0x141>(v);
    v += dpp_perm<0x140>(v);
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

struct Ffn2Args {
    const bf16_t* x16;
    const float* x32;
    const bf16_t* w1;
    const float* b1;
    const bf16_t* w2;
    const float* b2;
    const float* gamma;
    const float* beta;
    const int32_t* row_len;
    bf16_t* hid;
    uint16_t* bits;
    float* s_out;
    float* y32;
    bf16_t* y16;
    float* mean;
    float* rstd;
    int M, L, dff, Mp;
    float eps;
    asr_dropout_t drop;
    // PRE (appended: tools/gen_ffn_fwd.py reads the fields above by offset): the attention sub-layer's tail in front of this one.  x16 / x32
    // are then OUTPUTS of the launch's first phase (that sub-layer's y16 / y32) before they are this sub-layer's inputs.
    const bf16_t* pre_ctx16;
    const float* pre_res32;
    const bf16_t* pre_w;
    const float* pre_b;
    const float* pre_gamma;
    const float* pre_beta;
    float* pre_s_out;
    float* pre_mean;
    float* pre_rstd;
    float pre_eps;
    asr_dropout_t pre_drop;
};

typedef __attribute__((address_space(3))) void lds_void2;

// bias + dropout + residual + LayerNorm + row mask over 16 whole token rows read back from an LDS tile ([tokens][256] f32, 16-byte piece q
// of token t in slot q ^ (t & 7)); lane = 4 columns of the row (add_layernorm_fwd_kernel's arithmetic).  m0: the first row's token index.
// PREF: the fields of the phase in front (pre_*; its y32 / y16 are the launch's x32 / x16).  The arguments are read from the kernel-argument
// segment where they are used (`a`: a laundered pointer to it) - kept in scalar registers from the top of the kernel they would have to
// be spilled around the loop's block, which owns s20-s101.
typedef __attribute__((address_space(4))) const Ffn2Args* kernarg_ptr_t;
template <bool TRAIN, bool DROP, bool PREF>
__device__ __forceinline__ void ln_rows16(const unsigned char* tile, const f32x4 (&res)[16], __attribute__((address_space(4))) const Ffn2Args& a, int m0,
                                          int lane) {
    const int M = a.M, L = a.L;
    const int32_t* const row_len = a.row_len;
    const float* const bias = PREF ? a.pre_b : a.b2;
    const float* const gamma = PREF ? a.pre_gamma : a.gamma;
    const float* const beta = PREF ? a.pre_beta : a.beta;
    const float eps = PREF ? a.pre_eps : a.eps;
    float* const s_out = PREF ? a.pre_s_out : a.s_out;
    float* const y32 = PREF ? const_cast<float*>(a.x32) : a.y32;
    bf16_t* const y16 = PREF ? const_cast<bf16_t*>(a.x16) : a.y16;
    float* const mean_out = PREF ? a.pre_mean : a.mean;
    float* const rstd_out = PREF ? a.pre_rstd : a.rstd;
    asr_dropout_t drop_arg;      // (field by field: the argument block lives in the constant address space)
    if (PREF) { drop_arg.thr16 = a.pre_drop.thr16; drop_arg.key0 = a.pre_drop.key0; drop_arg.key1 = a.pre_drop.key1; drop_arg.salt = a.pre_drop.salt; }
    else { drop_arg.thr16 = a.drop.thr16; drop_arg.key0 = a.drop.key0; drop_arg.key1 = a.drop.key1; drop_arg.salt = a.drop.salt; }
    uint32_t keepmask = 0xffffu;
    const int m0c = m0 < M ? m0 : M - 1;
    const int b_first = m0c / L, t_first = m0c - b_first * L;
    if (row_len) {
        keepmask = 0;
        int bb = b_first, tt = t_first, len = row_len[bb];
        for (int tr = 0; tr < 16; ++tr) {
            keepmask |= (tt < len ? 1u : 0u) << tr;
            if (++tt == L) {
                tt = 0;
                if (m0 + tr + 1 < M) len = row_len[++bb];
            }
        }
    }
    const asr_dropout_t drop = drop_resolve(drop_arg);
    const float sc = drop_scale(drop);
    const f32x4 b2v = *reinterpret_cast<const f32x4*>(bias + 4 * lane);
    const f32x4 gm = *reinterpret_cast<const f32x4*>(gamma + 4 * lane), bt = *reinterpret_cast<const f32x4*>(beta + 4 * lane);
    const auto rss = __builtin_amdgcn_make_buffer_rsrc(s_out, 0, s_out ? (int)((int64_t)M * FD * 4) : 0, 0x00020000);
    const auto rsy = __builtin_amdgcn_make_buffer_rsrc(y32, 0, (int)((int64_t)M * FD * 4), 0x00020000);
    const auto rsz = __builtin_amdgcn_make_buffer_rsrc(y16, 0, y16 ? (int)((int64_t)M * FD * 2) : 0, 0x00020000);
    int bb = b_first, tt = t_first;
    uint32_t sub = DROP ? drop_subkey(drop, (uint32_t)bb) : 0u;
    float mean_l = 0.f, rstd_l = 0.f;
#pragma unroll
    for (int tr0 = 0; tr0 < 16; tr0 += 8) {
        f32x4 v[8];
        float part[8];
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tr = tr0 + j;       // (its swizzle key is tr & 7: m0 is a multiple of 16 inside the tile)
            const f32x4 y = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (j & 7)) << 4));
            f32x4 wv4 = y + b2v;
            if (DROP) {
                wv4 = drop4(drop, sub, (uint32_t)tt, FD >> 1, (uint32_t)(4 * lane), wv4, sc);
                if (++tt == L) {
                    tt = 0;
                    sub = drop_subkey(drop, (uint32_t)++bb);
                }
            }
            v[j] = wv4 + res[tr];
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
        for (int j = 0; j < 8; ++j) rstd[j] = 1.0f / sqrtf(wave_sum_dpp(part[j]) * (1.f / FD) + eps);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const int tr = tr0 + j, row = m0 + tr;
            const bool rv = row < M;
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
    if (TRAIN && lane < 16 && m0 + lane < M) {      // (plain stores: two buffer descriptors less in scalar registers beside the loop's block)
        if (mean_out) mean_out[m0 + lane] = mean_l;
        if (rstd_out) rstd_out[m0 + lane] = rstd_l;
    }
}

// PRE: the attention sub-layer's tail (attention.py:58-60: fc -> dropout -> + residual -> layer_norm, encoder.py:77's row mask) runs in
// front, on the same 128 tokens: wave (p, w) multiplies the pair's 32 context rows with output units 128 w .. + 127 of the [256][256]
// projection (all four 32-KiB chunk images resident), the halves meet in the pair's LDS tile, each wave normalises 16 rows and writes
// them as that sub-layer's outputs - x16 / x32 of the feed-forward phase, which reads them back.  One boundary between two fat one-round
// launches less per encoder layer (the first one's stores drained, its slowest workgroup's tail, the second prologue's exposed round trips):
// 10 us per layer in the training step, 23 in the inference forward; the bytes fetched are the pair's (profiles/r6/pmc_summary.txt).
// The phase in front (PRE, see the kernel): every index is derived here from a thread index of its own, and the arguments are read
// where they are used - nothing of this phase is meant to stay in registers across the loop's block.
template <bool TRAIN, bool DROP>
__device__ __forceinline__ void attn_tail_phase(unsigned char* smem, const Ffn2Args& a0) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    const int lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wv & 3, w = wv >> 2;
    const int mbase = blockIdx.x * FBM + 32 * p;
    const auto rsw = __builtin_amdgcn_make_buffer_rsrc(const_cast<bf16_t*>(a0.pre_w), 0, FD * FD * 2, 0x00020000);
#pragma unroll
    for (int c = 0; c < 4; ++c)
#pragma unroll
        for (int j = 0; j < 4; ++j) {      // chunk image c: piece pi = rows 2 pi, 2 pi + 1 (512 B each); slot pc of row u holds 16-byte chunk (pc & 16) | ((pc ^ u) & 15)
            const int pi = wv * 4 + j, u = 2 * pi + (lane >> 5), pc = lane & 31;
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rsw, (lds_void2*)(smem + c * 32768 + pi * 1024), 16,
                                                     (unsigned)(u * FD * 2 + ((pc & 16) | ((pc ^ u) & 15)) * 16), c * (FHC * FD * 2), 0, 0);
        }
    // wave (p, w): tokens 16 w .. + 15 of pair p against ALL 256 output units on v_mfma_f32_16x16x32_bf16 - the wave then holds
    // complete rows of exactly the 16 tokens it normalises (at 32 x 32 the pair would split the units and hold 64 + 64 + 64
    // registers of operand, accumulator and residual: more than a wave has at two per SIMD beside the loop's fixed 176 + 64)
    const int r16 = lane & 15, q4 = lane >> 4;
    const int m0 = mbase + 16 * w;
    const int mr = m0 + r16 < a0.M ? m0 + r16 : a0.M - 1;
    u32x4 xb[8];
    {
        const bf16_t* xr = a0.pre_ctx16 + (int64_t)mr * FD + 8 * q4;
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) xb[ks] = *reinterpret_cast<const u32x4*>(xr + 32 * ks);
    }
    // the residual rows of this wave's 16 rows (row layout: lane = 4 columns): requested with everything else the phase reads - one HBM
    // round trip for the weight images, the context rows and these (64 + 32 + 64 registers with the accumulators: inside a wave's 176)
    f32x4 res[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rowc = m0 + k < a0.M ? m0 + k : a0.M - 1;
        res[k] = *reinterpret_cast<const f32x4*>(a0.pre_res32 + (int64_t)rowc * FD + 4 * lane);
    }
    f32x4 acc[16];
#pragma unroll
    for (int j = 0; j < 16; ++j) acc[j] = f32x4{0.f, 0.f, 0.f, 0.f};
    // fragment of unit row u = 16 j + r16 (row u & 63 of chunk image u >> 6), k-step ks: the 16-byte piece pc = 4 ks + q4 of the row
    unsigned fa[8];
#pragma unroll
    for (int ks = 0; ks < 8; ++ks) {
        const int pc = 4 * ks + q4;
        fa[ks] = (unsigned)(r16 * 512 + (((pc & 16) | ((pc ^ r16) & 15)) << 4));
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
#pragma unroll
    for (int j = 0; j < 16; ++j) {      // (rows 16 j .. + 15: row & 15 = r16 for every j, so one address set serves all sixteen tiles)
        const unsigned char* wj = smem + (j >> 2) * 32768 + (j & 3) * (16 * 512);
#pragma unroll
        for (int ks = 0; ks < 8; ++ks) Mma<bf16_t>::run(*reinterpret_cast<const u32x4*>(wj + fa[ks]), xb[ks], acc[j]);
        __builtin_amdgcn_sched_barrier(0);
    }
    __syncthreads();      // every wave is past its fragment reads: the images become the waves' row tiles
    {
        // lane (r16, q4) holds units 16 j + 4 q4 .. + 3 of token r16: 16-byte piece 4 j + q4 of the row, slot piece ^ (token & 7)
        unsigned char* const trow = smem + p * 32768 + (16 * w + r16) * 1024;
#pragma unroll
        for (int j = 0; j < 16; ++j) *reinterpret_cast<f32x4*>(trow + (((4 * j + q4) ^ (r16 & 7)) << 4)) = acc[j];
    }
    // (wave-private rows: written and read back by this wave only, LDS operations of a wave complete in order)
    {
        kernarg_ptr_t pap = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
        asm volatile("" : "+s"(pap));
        // (lane and m0 through a barrier of their own: the row offsets of this call are those of the epilogue's call, and what the
        // compiler keeps of them across the loop's block - which owns v0-v175 - it parks in AGPRs beyond the block's 64)
        int lane_p = lane, m0_p = m0;
        asm volatile("" : "+v"(lane_p), "+s"(m0_p));
        ln_rows16<TRAIN, DROP, true>(smem + p * 32768 + (16 * w) * 1024, res, *pap, m0_p, lane_p);
    }
    // the rows are this launch's own inputs from here on: written (acknowledged by L2) before any wave of the workgroup reads them
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
}

template <bool TRAIN, bool DROP, bool PRE>
__global__ __launch_bounds__(512, 1) void ffn_fwd2_kernel(const Ffn2Args a0) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[SMEM2];
    if constexpr (PRE) attn_tail_phase<TRAIN, DROP>(smem, a0);
    const int tid = threadIdx.x, lane = tid & 63;
    const int wv = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int p = wv & 3, w = wv >> 2;
    const int r = lane & 31, h = lane >> 5;
    const int mbase = blockIdx.x * FBM + 32 * p;
    const int m = mbase + r;
    const int mc = m < a0.M ? m : a0.M - 1;
    const int dff = a0.dff, NC = dff / FHC;
    const unsigned smem0 = lds_addr_of(smem);
    {
        // ---- the loop's per-lane parameters (tools/gen_ffn_fwd.py: PARAMS), formulas of ffn.hip's images ----
        unsigned* const P = reinterpret_cast<unsigned*>(smem + W2RING) + tid;
        auto put = [&](int k, unsigned val) { P[k * 512] = val; };
        const int ur = swap23(r), u15 = ur & 15;
        // W1 chunk image [64 units][512 B]: 16-byte slot pc of row u holds chunk (pc & 16) | ((pc ^ u) & 15); this wave's units 32 w + ur
#pragma unroll
        for (int kk = 0; kk < 8; ++kk) put(FFN2_P_A1_0 + kk, smem0 + (unsigned)(w * 16384 + ur * 512 + (((2 * kk + h) ^ u15) << 4)));
        // W2 chunk image [256 d][128 B]: slot pc of row d holds chunk pc ^ ((d >> 1) & 7); this wave's rows 128 w + 32 ytl + r; k-step sg
        auto a2 = [&](int sg) { return smem0 + (unsigned)(W2RING + w * 16384 + r * 128 + (((2 * sg + h) ^ ((r >> 1) & 7)) << 4)); };
        put(FFN2_P_A2O_0, a2(2 * w));
        put(FFN2_P_A2O_1, a2(2 * w + 1));
        put(FFN2_P_A2P_0, a2(2 * (1 - w)));
        put(FFN2_P_A2P_1, a2(2 * (1 - w) + 1));
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            const int pi = wv * 4 + j;
            {
                const int u = 2 * pi + (lane >> 5), pc = lane & 31;
                put(FFN2_P_OFF1_0 + j, (unsigned)(u * FD * 2 + ((pc & 16) | ((pc ^ u) & 15)) * 16));
            }
            {
                const int d = 8 * pi + (lane >> 3), pc = lane & 7;
                put(FFN2_P_OFF2_0 + j, (unsigned)(d * dff * 2 + (pc ^ ((d >> 1) & 7)) * 16));
            }
        }
        // the pair's H tile [32 tok][128 B], 16-byte slot s of token t at s ^ (t & 7): this lane writes units 32 w + 16 g + 8 h .. + 8 of
        // its token (slot 4 w + 2 g + h), reads the partner's, and reads back whole rows (tokens 16 w + 8 ps + lane / 8) for the stores
        const unsigned hx = smem0 + (unsigned)(HXRING + p * 4096);
#pragma unroll
        for (int g = 0; g < 2; ++g) {
            put(FFN2_P_HXW_0 + g, hx + (unsigned)(r * 128 + (((4 * w + 2 * g + h) ^ (r & 7)) << 4)));
            put(FFN2_P_HXP_0 + g, hx + (unsigned)(r * 128 + (((4 * (1 - w) + 2 * g + h) ^ (r & 7)) << 4)));
            const int tk = 16 * w + 8 * g + (lane >> 3), pc = lane & 7;
            put(FFN2_P_HXR_0 + g, hx + (unsigned)(tk * 128 + ((pc ^ (tk & 7)) << 4)));
            const int mt = mbase + tk;
            put(FFN2_P_HOFF_0 + g, (TRAIN && mt < a0.M) ? (unsigned)mt * (unsigned)dff * 2u + 16u * pc : 0x80000000u);
        }
        put(FFN2_P_BOFF, (TRAIN && m < a0.M) ? ((unsigned)h * a0.Mp + m) * 2u : 0x80000000u);
        put(FFN2_P_XOFF, (unsigned)mc * (FD * 2u) + 16u * h);
    }
    __syncthreads();
    {
        // the block reads the kernel's arguments itself (tools/gen_ffn_fwd.py: KA_*), so that its only inputs are three scalars
        static_assert(offsetof(Ffn2Args, x16) == 0 && offsetof(Ffn2Args, w1) == 16 && offsetof(Ffn2Args, b1) == 24 && offsetof(Ffn2Args, w2) == 32 &&
                          offsetof(Ffn2Args, hid) == 72 && offsetof(Ffn2Args, bits) == 80 && offsetof(Ffn2Args, M) == 128 &&
                          offsetof(Ffn2Args, dff) == 136 && offsetof(Ffn2Args, Mp) == 140, "tools/gen_ffn_fwd.py reads Ffn2Args by offset");
        const uint64_t ka = (uint64_t)__builtin_amdgcn_kernarg_segment_ptr();
        if constexpr (TRAIN) asm volatile(FFN2_FWD_ASM_TRAIN : : [ka] "s"(ka), [wv] "s"(wv), [smem0] "s"(smem0) : FFN2_FWD_ASM_CLOBBERS);
        else asm volatile(FFN2_FWD_ASM_EVAL : : [ka] "s"(ka), [wv] "s"(wv), [smem0] "s"(smem0) : FFN2_FWD_ASM_CLOBBERS);
    }
    // ---- epilogue: v = dropout(Y + b2) + x, LayerNorm, row mask; this wave's 16 rows of the pair's tile [32 tok][256] f32 ----------------
    // (the arguments are read again from the kernel-argument segment: kept in scalar registers across the block they would have to be
    // spilled - the block owns s20-s101)
    kernarg_ptr_t eap = (kernarg_ptr_t)__builtin_amdgcn_kernarg_segment_ptr();
    asm volatile("" : "+s"(eap));
    __attribute__((address_space(4))) const Ffn2Args& a = *eap;
    const int m0 = mbase + 16 * w;
    f32x4 res[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int rowc = m0 + k < a.M ? m0 + k : a.M - 1;
        res[k] = *reinterpret_cast<const f32x4*>(a.x32 + (int64_t)rowc * FD + 4 * lane);
    }
    __syncthreads();
    ln_rows16<TRAIN, DROP, false>(smem + p * 32768 + (16 * w) * 1024, res, a, m0, lane);
}

}  // namespace

// asr_ffn_fwd's launch (ffn.hip checks the arguments and calls here); pre != nullptr: asr_attn_ffn_fwd's
int asr_ffn_fwd2_launch(hipStream_t stream, const void* x16, const float* x32, const void* w1, const float* b1, const void* w2, const float* b2,
                        const float* gamma, const float* beta, const int32_t* row_len, void* hid_out, void* bits_out, float* s_out, float* y32,
                        void* y16, float* mean_out, float* rstd_out, int M, int L, int d_ff, float eps, asr_dropout_t drop_x, const asr_ffn2_pre_t* pre) {
    Ffn2Args a{(const bf16_t*)x16, x32, (const bf16_t*)w1, b1, (const bf16_t*)w2, b2, gamma, beta, row_len, (bf16_t*)hid_out,
               (uint16_t*)bits_out, s_out, y32, (bf16_t*)y16, mean_out, rstd_out, M, L, d_ff, (M + FBM - 1) / FBM * FBM, eps, drop_x,
               nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, nullptr, 0.f, asr_dropout_t{}};
    if (pre) {
        a.pre_ctx16 = (const bf16_t*)pre->ctx16; a.pre_res32 = pre->res32; a.pre_w = (const bf16_t*)pre->w; a.pre_b = pre->bias;
        a.pre_gamma = pre->gamma; a.pre_beta = pre->beta; a.pre_s_out = pre->s_out; a.pre_mean = pre->mean_out; a.pre_rstd = pre->rstd_out;
        a.pre_eps = pre->eps; a.pre_drop = pre->drop_x;
    }
    const dim3 grid((M + FBM - 1) / FBM), block(512);
    const bool dr = drop_x.thr16 != 0;
#define FFN2_GO(T, D, P) hipLaunchKernelGGL((ffn_fwd2_kernel<T, D, P>), grid, block, 0, stream, a)
    if (pre) {
        if (hid_out && dr) FFN2_GO(true, true, true);
        else if (hid_out) FFN2_GO(true, false, true);
        else if (dr) FFN2_GO(false, true, true);
        else FFN2_GO(false, false, true);
    } else {
        if (hid_out && dr) FFN2_GO(true, true, false);
        else if (hid_out) FFN2_GO(true, false, false);
        else if (dr) FFN2_GO(false, true, false);
        else FFN2_GO(false, false, false);
    }
#undef FFN2_GO
    ASR_LAUNCH_CHECK(pre ? "asr_attn_ffn_fwd" : "asr_ffn_fwd");
    return 0;
}
