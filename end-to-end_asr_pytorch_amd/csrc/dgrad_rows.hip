// Data gradient of an nn.Linear whose INPUT is d_model = 256 wide, at encoder size, as a row-block kernel:
//   dX[M, 256] = dY[M, K] . W[K, 256]  (+ addend)          autograd of y = x W^T (attention.py:43-49 w_qs / w_ks / w_vs, :58 fc; the
//   decoder's cross-attention K / V projections over the encoder output, decoder.py:216-230), W as nn.Linear stores it: [out = K][in = 256]
// A tiled GEMM sees a skinny problem here (N = 256, K = 256 ... 3072): 500 tiles with a short reduction each, every one of them
// mostly prologue and epilogue - 29.5 us for K = 768 where the bytes (dY in, addend in, dX out: 115 MB) take 18.  This kernel is
// csrc/ffn.hip's data-gradient kernel without its first product: a workgroup owns 128 rows of dX complete (4 waves x 32 rows, the
// 256 x 32 accumulator of a wave = 128 registers), W streams through LDS in 64-row chunks exactly as W1 does there (row-major image,
// transposing reads, same swizzle), and the rows' dY chunk [128][64] rides in the same ring stage (16 KiB, 16-byte slot ^ (row & 7):
// the B operand of a lane is one conflict-free ds_read_b128).  Three stages, requested two chunks ahead, one barrier per chunk.
// Because the rows are complete, the epilogue can be more than a store:
//   OUT_F32   dX (+ addend) as f32 rows                         (the encoder-output gradient of the decoder's cross K / V: addend = what
//                                                                is there already)
//   OUT_BF16  dX as bf16 rows                                   (fc's gradient wrt the attention context: the operand of attention_bwd)
//   OUT_LNB   dX + addend never leaves: it is the dy of the LayerNorm that produced x (module.py:52 / encoder.py:48-50), and the
//             launch writes what asr_add_layernorm_bwd writes - ds, ds16, dgamma / dbeta / dbias column sums (same arithmetic per row as
//             that kernel and as asr_ffn_bwd_ln; unlike there the fold pays: this kernel is bound by its bytes, and the fold removes
//             65 MB of them - dx out, dy in - together with a launch)
#include <hip/hip_runtime.h>

#include "asr_common.h"

namespace {

constexpr int RBM = 128;               // rows per workgroup
constexpr int RKC = 64;                // reduction rows of W per chunk
constexpr int RD = 256;                // d_model
constexpr int RWBUF = RKC * RD * 2;    // 32 KiB: [64 k][256 d] bf16, 512-byte rows
constexpr int RYBUF = RBM * RKC * 2;   // 16 KiB: [128 rows][64 k] bf16, 128-byte rows
constexpr int RSTAGE = RWBUF + RYBUF;
constexpr int RNST = 3;

enum { OUT_F32 = 0, OUT_BF16 = 1, OUT_LNB = 2 };

struct RowsArgs {
    const bf16_t* dy;
    const bf16_t* w;
    const float* addend;      // f32 [M, 256] or null
    float* out32;
    bf16_t* out16;
    int M, K;
    long long ldy;            // elements between rows of dY
    // OUT_LNB
    const float* ln_s;
    const float* ln_mean;
    const float* ln_rstd;
    const float* ln_gamma;
    const float* ln_beta;     // non-null: ln_s holds the LayerNorm's output y, x^ = (y - beta) / gamma
    const int32_t* row_len;
    int L;
    float* dgamma;
    float* dbeta;
    float* dbias;
    asr_dropout_t drop_x;
};

typedef __attribute__((ext_vector_type(4))) short s16x4_t;
__device__ __forceinline__ bf16x8 tr_pair_rows(const unsigned char* p, int second_off) {
    const s16x4_t v0 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)p);
    const s16x4_t v1 = __builtin_amdgcn_ds_read_tr16_b64_v4i16((__attribute__((address_space(3))) s16x4_t*)(p + second_off));
    const u32x2 a = __builtin_bit_cast(u32x2, v0), b = __builtin_bit_cast(u32x2, v1);
    return __builtin_bit_cast(bf16x8, u32x4{a[0], a[1], b[0], b[1]});
}

template <int CTRL> __device__ __forceinline__ float dpp_perm_rows(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xf, 0xf, true));
}
__device__ __forceinline__ float wave_sum_rows(float v) {      // (csrc/ffn.hip's wave_sum_dpp)
    v += dpp_perm_rows<0xB1>(v);
    v += dpp_perm_rows<0x4E>(v);
    v += dpp_perm_rows<0x141>(v);
    v += dpp_perm_rows<0x140>(v);
    const int iv = __builtin_bit_cast(int, v);
    return (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 0)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 16))) +
           (__builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 32)) + __builtin_bit_cast(float, __builtin_amdgcn_readlane(iv, 48)));
}

template <int OUT, bool ADD>
__global__ __launch_bounds__(256, 1) void dgrad_rows_kernel(const RowsArgs a) {
    __shared__ __attribute__((aligned(16))) unsigned char smem[RNST * RSTAGE];
    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 31, h = lane >> 5;
    const int m0 = blockIdx.x * RBM + wave * 32;
    const int NC = a.K / RKC;

    // ---- staging: W chunk pieces (8 per wave, 1 KiB = two 512-byte rows) and dY chunk pieces (4 per wave, 1 KiB = eight 128-byte rows)
    const u32x4 rsw = rsrc_words(a.w, (unsigned)((int64_t)a.K * RD * 2));
    const u32x4 rsy = rsrc_words(a.dy, (unsigned)(((int64_t)(a.M - 1) * a.ldy + a.K) * 2));
    unsigned offw[8], offy[4];
#pragma unroll
    for (int k = 0; k < 8; ++k) {
        const int p = wave * 8 + k;
        const int u = 2 * p + (lane >> 5), pc = lane & 31;
        offw[k] = (unsigned)(u * RD * 2 + ((pc ^ ((u & 3) << 2)) << 4));
    }
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        const int p = wave * 4 + k;
        const int row = 8 * p + (lane >> 3), pc = lane & 7;
        const int mrow = blockIdx.x * RBM + row < a.M ? blockIdx.x * RBM + row : a.M - 1;
        offy[k] = (unsigned)((int64_t)mrow * a.ldy * 2 + ((pc ^ (row & 7)) << 4));
    }
    auto dma_chunk = [&](int stage, int chunk) {
        unsigned char* const ws = smem + stage * RSTAGE;
#pragma unroll
        for (int j = 0; j < 8; ++j) dma16_asm(rsw, offw[j], chunk * (RKC * RD * 2), lds_addr_of(ws + (wave * 8 + j) * 1024));
#pragma unroll
        for (int j = 0; j < 4; ++j) dma16_asm(rsy, offy[j], chunk * (RKC * 2), lds_addr_of(ws + RWBUF + (wave * 4 + j) * 1024));
    };

    // transposed-read addresses of the W image (csrc/ffn.hip, second product of the data gradient)
    const int q = (lane & 15) >> 2, pq = lane & 3, g1 = (lane >> 4) & 1;
    unsigned a2[4];
#pragma unroll
    for (int v = 0; v < 4; ++v) a2[v] = (unsigned)((8 * h + q) * 512 + (4 * (v ^ q) + 2 * g1 + (pq >> 1)) * 16 + (pq & 1) * 8);
    auto fragw = [&](const unsigned char* w, int k) {      // MFMA k of a chunk: k-step k >> 3 (16 rows of the image), row tile k & 7 of dX^T
        return tr_pair_rows(w + a2[k & 3] + (k >> 3) * 8192 + ((k & 7) >> 2) * 256, 2048);
    };
    const unsigned yrd = (unsigned)((wave * 32 + r) * 128);

    f32x16 Y[8];
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int j = 0; j < 16; ++j) Y[t][j] = 0.f;

    // The epilogue's inputs - the addend rows and, for the LayerNorm fold, the rows of the pre-norm sum (row layout: lane = 4 columns
    // of a row) - are requested FIRST, before the weights: they land while the loop runs.  Every workgroup of the single round is in
    // the same phase, so whatever the epilogue still has to fetch is fetched by a chip that is doing nothing else (requested at the
    // end of the loop instead: 64 us for K = 768 where the tiled GEMM took 48).  Half the register file is free for them at one wave
    // per SIMD.  They are older than every LDS-DMA request, so the counted waits below cover them from the first iteration on.
    f32x4 res[32], srow[32];
    if (ADD) {
#pragma unroll
        for (int k = 0; k < 32; ++k) {
            const int rowc = m0 + k < a.M ? m0 + k : a.M - 1;
            res[k] = *reinterpret_cast<const f32x4*>(a.addend + (int64_t)rowc * RD + 4 * lane);
        }
    }
    if (OUT == OUT_LNB) {
#pragma unroll
        for (int tr = 0; tr < 32; ++tr) {
            const int rowc = m0 + tr < a.M ? m0 + tr : a.M - 1;
            srow[tr] = *reinterpret_cast<const f32x4*>(a.ln_s + (int64_t)rowc * RD + 4 * lane);
        }
    }
    dma_chunk(0, 0);
    dma_chunk(1, NC > 1 ? 1 : 0);
    for (int i = 0; i < NC; ++i) {
        // chunk i landed (this wave's part: everything older than the 12 requests of chunk i + 1), then everybody's
        asm volatile("s_waitcnt vmcnt(12)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        {   // chunk i + 2 into the stage chunk i - 1 was read from (past the end: the last chunk again - the count of requests in
            // flight stays what the wait above assumes)
            const int nx = i + 2 < NC ? i + 2 : NC - 1;
            dma_chunk((i + 2) % RNST, nx);
        }
        const unsigned char* ws = smem + (i % RNST) * RSTAGE;
        const unsigned char* ys = ws + RWBUF;
        bf16x8 B[4];
#pragma unroll
        for (int ks = 0; ks < 4; ++ks) B[ks] = *reinterpret_cast<const bf16x8*>(ys + yrd + (((2 * ks + h) ^ (r & 7)) << 4));
#pragma unroll
        for (int k = 0; k < 32; ++k) Y[k & 7] = __builtin_amdgcn_mfma_f32_32x32x16_bf16(fragw(ws, k), B[k >> 3], Y[k & 7], 0, 0, 0);
    }
    asm volatile("s_waitcnt vmcnt(0) lgkmcnt(0)" ::: "memory");      // (the surplus requests of the last two iterations: nothing may land later)
    __syncthreads();

    // ---- epilogue through the wave's 32-KiB LDS tile (csrc/ffn.hip): loads and stores are whole rows ---------------------------------
    unsigned char* const tile = smem + wave * 32768;
#pragma unroll
    for (int t = 0; t < 8; ++t)
#pragma unroll
        for (int g = 0; g < 4; ++g)
            *reinterpret_cast<f32x4*>(tile + r * 1024 + (((8 * t + 2 * g + h) ^ (r & 7)) << 4)) =
                f32x4{Y[t][4 * g], Y[t][4 * g + 1], Y[t][4 * g + 2], Y[t][4 * g + 3]};
    if constexpr (OUT == OUT_F32) {
        const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out32, 0, (int)((int64_t)a.M * RD * 4), 0x00020000);
#pragma unroll
        for (int tr = 0; tr < 32; ++tr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (tr & 7)) << 4));
            if (ADD) v += res[tr];
            const int row = m0 + tr;
            __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, v), rso, row < a.M ? (unsigned)row * (RD * 4u) + 16u * lane : 0x80000000u, 0, 0);
        }
    } else if constexpr (OUT == OUT_BF16) {
        const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out16, 0, (int)((int64_t)a.M * RD * 2), 0x00020000);
#pragma unroll
        for (int tr = 0; tr < 32; ++tr) {
            f32x4 v = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (tr & 7)) << 4));
            if (ADD) v += res[tr];
            const int row = m0 + tr;
            const bf16x4 ob = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
            __builtin_amdgcn_raw_buffer_store_b64(__builtin_bit_cast(u32x2, ob), rso, row < a.M ? (unsigned)row * (RD * 2u) + 8u * lane : 0x80000000u, 0, 0);
        }
    } else {
        // the backward of the LayerNorm whose output the projection read: add_layernorm_bwd_kernel's arithmetic on rows that never went
        // to memory as dx:  d = dX + addend (0 for masked rows);  xh = (s - mean) * rstd;  g = d * gamma;
        //   ds = (g - mean(g) - xh * mean(g * xh)) * rstd;   dgamma += d * xh, dbeta += d, dbias += dropout_x(ds)   (column sums)
        const int myrow = m0 + r < a.M ? m0 + r : a.M - 1;                    // lane (and lane + 32) keep row r's statistics
        const float mu_l = a.ln_mean ? a.ln_mean[myrow] : 0.f, rs_l = a.ln_rstd[myrow];
        const int b_l = myrow / a.L, t_l = myrow - b_l * a.L;
        const int keep_l = (m0 + r < a.M && t_l < (a.row_len ? a.row_len[b_l] : a.L)) ? 1 : 0;
        const f32x4 gam = *reinterpret_cast<const f32x4*>(a.ln_gamma + 4 * lane);
        const f32x4 betav = a.ln_beta ? *reinterpret_cast<const f32x4*>(a.ln_beta + 4 * lane) : f32x4{0, 0, 0, 0};
        const asr_dropout_t drop = drop_resolve(a.drop_x);
        const float scx = drop_scale(drop);
        const auto rso = __builtin_amdgcn_make_buffer_rsrc(a.out32, 0, (int)((int64_t)a.M * RD * 4), 0x00020000);
        const auto rsh16 = __builtin_amdgcn_make_buffer_rsrc(a.out16, 0, (int)((int64_t)a.M * RD * 2), 0x00020000);
        f32x4 ag = {0, 0, 0, 0}, ab = {0, 0, 0, 0}, as = {0, 0, 0, 0};
        const int m0c = m0 < a.M ? m0 : a.M - 1;
        int bb = m0c / a.L, tt = m0c - bb * a.L;                               // (utterance, frame) of row m0 + tr, kept in step
        constexpr int RW = 4;                                                 // rows in flight together: their reductions interleave
#pragma unroll
        for (int tp = 0; tp < 32; tp += RW) {
            f32x4 d[RW], xh[RW], g[RW];
            float s1[RW], s2[RW], rs[RW];
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int tr = tp + u;
                f32x4 y = *reinterpret_cast<const f32x4*>(tile + tr * 1024 + ((lane ^ (tr & 7)) << 4));
                if (ADD) y += res[tr];
                const float mu = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, mu_l), tr));
                rs[u] = __builtin_bit_cast(float, __builtin_amdgcn_readlane(__builtin_bit_cast(int, rs_l), tr));
                const bool keep = __builtin_amdgcn_readlane(keep_l, tr) != 0;
                d[u] = keep ? y : f32x4{0, 0, 0, 0};
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
                s1[u] = wave_sum_rows(s1[u]);
                s2[u] = wave_sum_rows(s2[u]);
            }
#pragma unroll
            for (int u = 0; u < RW; ++u) {
                const int row = m0 + tp + u;
                const float m1 = s1[u] * (1.f / RD), m2 = s2[u] * (1.f / RD);
                f32x4 o = (g[u] - m1 - xh[u] * m2) * rs[u];
                const unsigned off = row < a.M ? (unsigned)row * RD : 0x20000000u;      // (in elements; past the buffer for a row that is not there)
                __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(u32x4, o), rso, off * 4u + 16u * lane, 0, 0);     // gradient wrt the residual
                if (drop.thr16)                                                         // gradient wrt the normalised projection's output
                    o = drop4(drop, drop_subkey(drop, (uint32_t)bb), (uint32_t)tt, RD >> 1, 4 * lane, o, scx);
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
            red[(0 * 4 + wave) * RD + 4 * lane + k] = ag[k];
            red[(1 * 4 + wave) * RD + 4 * lane + k] = ab[k];
            red[(2 * 4 + wave) * RD + 4 * lane + k] = as[k];
        }
        __syncthreads();
#pragma unroll
        for (int which = 0; which < 3; ++which) {
            float* dst = which == 0 ? a.dgamma : (which == 1 ? a.dbeta : a.dbias);
            if (!dst) continue;
            const float v = (red[(which * 4 + 0) * RD + tid] + red[(which * 4 + 1) * RD + tid]) + (red[(which * 4 + 2) * RD + tid] + red[(which * 4 + 3) * RD + tid]);
            atomicAdd(dst + tid, v);
        }
    }
}

template <int OUT>
void launch_rows(hipStream_t s, const RowsArgs& a) {
    const dim3 grid((a.M + RBM - 1) / RBM), block(256);
    if (a.addend) hipLaunchKernelGGL((dgrad_rows_kernel<OUT, true>), grid, block, 0, s, a);
    else hipLaunchKernelGGL((dgrad_rows_kernel<OUT, false>), grid, block, 0, s, a);
}

int check_common(const char* who, const void* dy, long long ldy, const void* w, int M, int K, int d_model) {
    ASR_REQUIRE(d_model == RD, ASR_ERR_UNSUPPORTED, "%s: d_model = %d (the row-block data gradient is built for 256)", who, d_model);
    ASR_REQUIRE(K >= RKC && K % RKC == 0, ASR_ERR_UNSUPPORTED, "%s: K = %d (a multiple of 64)", who, K);
    ASR_REQUIRE(M > 0 && ldy >= K && ldy % 8 == 0 && ((int64_t)(M - 1) * ldy + K) * 2 < (1ll << 32), ASR_ERR_ARG, "%s: bad M / row stride", who);
    ASR_REQUIRE(dy && w, ASR_ERR_ARG, "%s: null argument", who);
    ASR_REQUIRE(asr_aligned(dy, 16) && asr_aligned(w, 16), ASR_ERR_ALIGN, "%s: 16-byte aligned operands required", who);
    return 0;
}

}  // namespace

extern "C" int asr_dgrad_rows(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, void* out, int out_dtype, int M,
                              int K, int d_model) {
    if (int rc = check_common("asr_dgrad_rows", dy, ldy, w, M, K, d_model)) return rc;
    ASR_REQUIRE(out && (out_dtype == ASR_F32 || out_dtype == ASR_BF16), ASR_ERR_ARG, "asr_dgrad_rows: bad output");
    ASR_REQUIRE(asr_aligned(out, 16) && (!addend || asr_aligned(addend, 16)), ASR_ERR_ALIGN, "asr_dgrad_rows: 16-byte aligned buffers required");
    RowsArgs a{};
    a.dy = (const bf16_t*)dy; a.w = (const bf16_t*)w; a.addend = addend; a.M = M; a.K = K; a.ldy = ldy;
    if (out_dtype == ASR_F32) { a.out32 = (float*)out; launch_rows<OUT_F32>((hipStream_t)stream, a); }
    else { a.out16 = (bf16_t*)out; launch_rows<OUT_BF16>((hipStream_t)stream, a); }
    ASR_LAUNCH_CHECK("asr_dgrad_rows");
    return 0;
}

extern "C" int asr_dgrad_rows_ln(void* stream, const void* dy, int64_t ldy, const void* w, const float* addend, int B, int L, int K,
                                 int d_model, const float* ln_s, const float* ln_mean, const float* ln_rstd, const float* ln_gamma,
                                 const float* ln_beta, const int32_t* row_len, float* ds_out, void* ds16_out, float* dgamma, float* dbeta, float* dbias,
                                 asr_dropout_t drop_x) {
    ASR_REQUIRE(B > 0 && L > 0, ASR_ERR_ARG, "asr_dgrad_rows_ln: bad B / L");
    const int M = B * L;
    if (int rc = check_common("asr_dgrad_rows_ln", dy, ldy, w, M, K, d_model)) return rc;
    ASR_REQUIRE(ln_s && (ln_mean || ln_beta) && ln_rstd && ln_gamma && ds_out && ds16_out && dgamma && dbeta, ASR_ERR_ARG, "asr_dgrad_rows_ln: null argument");
    ASR_REQUIRE(asr_aligned(ln_s, 16) && asr_aligned(ln_gamma, 16) && asr_aligned(ds_out, 16) && asr_aligned(ds16_out, 16) &&
                    (!addend || asr_aligned(addend, 16)), ASR_ERR_ALIGN, "asr_dgrad_rows_ln: 16-byte aligned buffers required");
    RowsArgs a{};
    a.dy = (const bf16_t*)dy; a.w = (const bf16_t*)w; a.addend = addend; a.M = M; a.K = K; a.ldy = ldy;
    a.out32 = ds_out; a.out16 = (bf16_t*)ds16_out;
    a.ln_s = ln_s; a.ln_mean = ln_mean; a.ln_rstd = ln_rstd; a.ln_gamma = ln_gamma; a.ln_beta = ln_beta; a.row_len = row_len; a.L = L;
    a.dgamma = dgamma; a.dbeta = dbeta; a.dbias = dbias; a.drop_x = drop_x;
    launch_rows<OUT_LNB>((hipStream_t)stream, a);
    ASR_LAUNCH_CHECK("asr_dgrad_rows_ln");
    return 0;
}
