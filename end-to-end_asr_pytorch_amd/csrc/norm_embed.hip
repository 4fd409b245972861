// HBM-bound row kernels: residual-add + LayerNorm (+PE, +length mask), embedding + PE, casts, row masks,
// assigner tail.  One 64-lane wave owns one row (D <= 1024), 16-byte accesses, reductions by cross-lane shuffles.
#include "asr_common.h"

namespace {

constexpr int LN_MAXJ = 4;  // D <= 1024

__global__ __launch_bounds__(256) void add_layernorm_fwd_kernel(const float* x, const float* __restrict__ res,
                                                                const float* __restrict__ gamma, const float* __restrict__ beta,
                                                                const float* __restrict__ pe, const int32_t* __restrict__ row_len,
                                                                float* __restrict__ y32, bf16_t* __restrict__ y16,
                                                                float* __restrict__ mean_out, float* __restrict__ rstd_out,
                                                                float* s_out, int M, int L, int D, float eps,
                                                                asr_dropout_t drop_x_in, asr_dropout_t drop_y_in) {
    const asr_dropout_t drop_x = drop_resolve(drop_x_in), drop_y = drop_resolve(drop_y_in);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = (int)(row / L), t = (int)(row - (int64_t)b * L);
    const float* xr = x + row * D;
    f32x4 v[LN_MAXJ];
    float s = 0.f;
    const uint32_t subx = drop_x.thr16 ? drop_subkey(drop_x, b) : 0u, suby = drop_y.thr16 ? drop_subkey(drop_y, b) : 0u;
    const float scx = drop_scale(drop_x), scy = drop_scale(drop_y);
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
            v[j] = *reinterpret_cast<const f32x4*>(xr + c);
            if (drop_x.thr16) v[j] = drop4(drop_x, subx, t, D >> 1, c, v[j], scx);
            if (res) v[j] += *reinterpret_cast<const f32x4*>(res + row * D + c);
            if (s_out) *reinterpret_cast<f32x4*>(s_out + row * D + c) = v[j];   // pre-norm sum for the backward (may alias x)
            s += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
        }
    }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
            const f32x4 d = v[j] - mean;
            q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
        }
    }
    const float var = wave_sum(q) / (float)D;
    const float rstd = 1.0f / sqrtf(var + eps);
    if (lane == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    const bool keep = row_len ? (t < row_len[b]) : true;
#pragma unroll
    for (int j = 0; j < LN_MAXJ; ++j) {
        const int c = lane * 4 + 256 * j;
        if (c < D) {
            const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
            const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
            f32x4 o = (v[j] - mean) * rstd * g + bt;
            if (pe) o += *reinterpret_cast<const f32x4*>(pe + (int64_t)t * D + c);
            if (drop_y.thr16) o = drop4(drop_y, suby, t, D >> 1, c, o, scy);
            if (!keep) o = f32x4{0, 0, 0, 0};
            *reinterpret_cast<f32x4*>(y32 + row * D + c) = o;
            if (y16) {
                bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                *reinterpret_cast<bf16x4*>(y16 + row * D + c) = ob;
            }
        }
    }
}

__global__ __launch_bounds__(256) void embed_pe_fwd_kernel(const int64_t* __restrict__ ids, const float* __restrict__ emb,
                                                           const float* __restrict__ pe, float* __restrict__ y32,
                                                           bf16_t* __restrict__ y16, int M, int U, int D, int V,
                                                           asr_dropout_t drop_in) {
    const asr_dropout_t drop = drop_resolve(drop_in);
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int u = (int)(row % U);
    int64_t id = ids[row];
    id = id < 0 ? 0 : (id >= V ? V - 1 : id);  // torch would raise; clamp keeps the kernel memory-safe
    const uint32_t sub = drop.thr16 ? drop_subkey(drop, (uint32_t)(row / U)) : 0u;
    const float sc = drop_scale(drop);
    for (int c = lane * 4; c < D; c += 256) {
        f32x4 o = *reinterpret_cast<const f32x4*>(emb + id * D + c) + *reinterpret_cast<const f32x4*>(pe + (int64_t)u * D + c);
        if (drop.thr16) o = drop4(drop, sub, u, D >> 1, c, o, sc);
        *reinterpret_cast<f32x4*>(y32 + row * D + c) = o;
        if (y16) {
            bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(y16 + row * D + c) = ob;
        }
    }
}

// generic dropout over f32 [N0,N1,N2]: one thread per element pair (n2 even/odd halves of one random word)
__global__ __launch_bounds__(256) void dropout_apply_kernel(const float* x, float* y, int N1, int N2, int64_t pairs_per_n0,
                                                            int64_t total_pairs, asr_dropout_t drop_in) {
    const asr_dropout_t drop = drop_resolve(drop_in);
    const int n2h = (N2 + 1) >> 1;
    const float sc = drop_scale(drop);
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total_pairs; i += stride) {
        const int64_t n0 = i / pairs_per_n0;
        const uint32_t pair = (uint32_t)(i - n0 * pairs_per_n0);
        const uint32_t n1 = pair / n2h, ph = pair - n1 * n2h;
        const uint32_t w = drop_word(drop, drop_subkey(drop, (uint32_t)n0), pair);
        const int64_t e = (n0 * N1 + n1) * N2 + 2 * ph;
        y[e] = drop_keep_lo(drop, w) ? x[e] * sc : 0.f;
        if (2 * ph + 1 < (uint32_t)N2) y[e + 1] = drop_keep_hi(drop, w) ? x[e + 1] * sc : 0.f;
    }
}

__global__ __launch_bounds__(256) void cast_f32_bf16_kernel(const float* __restrict__ x, bf16_t* __restrict__ y, int64_t n4,
                                                            int64_t n) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n4; i += stride) {
        const f32x4 v = *reinterpret_cast<const f32x4*>(x + i * 4);
        bf16x4 o = {(bf16_t)v[0], (bf16_t)v[1], (bf16_t)v[2], (bf16_t)v[3]};
        *reinterpret_cast<bf16x4*>(y + i * 4) = o;
    }
    if (blockIdx.x == 0 && threadIdx.x < (n - n4 * 4)) y[n4 * 4 + threadIdx.x] = (bf16_t)x[n4 * 4 + threadIdx.x];
}

__global__ __launch_bounds__(256) void mask_rows_kernel(float* __restrict__ x, const int32_t* __restrict__ len, int L, int V,
                                                        int64_t ld) {
    const int row = blockIdx.x;  // b*L + t
    const int b = row / L, t = row - b * L;
    if (t < len[b]) return;
    float* p = x + (int64_t)row * ld;
    for (int c = threadIdx.x; c < V; c += blockDim.x) p[c] = 0.f;
}

__global__ __launch_bounds__(256) void assigner_tail_kernel(const float* __restrict__ x, const float* __restrict__ w,
                                                            const float* __restrict__ bias, const int32_t* __restrict__ len,
                                                            int M, int L, int Dh, float* __restrict__ alpha) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    float s = 0.f;
    for (int c = lane; c < Dh; c += 64) s = fmaf(x[row * Dh + c], w[c], s);
    s = wave_sum(s) + bias[0];
    const int b = (int)(row / L), t = (int)(row - (int64_t)b * L);
    const float a = 1.0f / (1.0f + expf(-s));
    if (lane == 0) alpha[row] = (t < len[b]) ? a : 0.f;
}

// Decoder.preprocess (src/transformer/decoder.py:42-58) as one launch: one wave per utterance strips the pad (0) entries of its
// target row in order (ballot + prefix count), writes <sos> + tokens + 0-padding and tokens + <eos> + 0-padding, both [B, W], and the
// row's lengths.  Tokens that do not fit (more than W - 1 non-pad entries) set *overflow.
__global__ __launch_bounds__(256) void decoder_targets_kernel(const int64_t* __restrict__ targets, int64_t* __restrict__ ys_in,
                                                              int64_t* __restrict__ ys_out, int32_t* __restrict__ in_len,
                                                              int64_t* __restrict__ n_out, int32_t* __restrict__ overflow, int B, int U,
                                                              int W, int64_t sos, int64_t eos) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t* row = targets + (int64_t)b * U;
    int64_t* yi = ys_in + (int64_t)b * W;
    int64_t* yo = ys_out + (int64_t)b * W;
    int n = 0;
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const int64_t v = u < U ? row[u] : 0;
        const unsigned long long keep = __ballot(v != 0);
        const int pos = n + __popcll(keep & ((1ull << lane) - 1ull));
        if (v != 0 && pos < W - 1) {
            yi[1 + pos] = v;
            yo[pos] = v;
        }
        n += __popcll(keep);
    }
    const int nc = n < W - 1 ? n : W - 1;
    if (lane == 0) {
        yi[0] = sos;
        yo[nc] = eos;
        if (in_len) in_len[b] = (sos > 0 ? 1 : 0) + nc;      // what (ys_in > 0).sum(1) of decoder.py:83's non_pad_mask input counts (token ids are > 0)
        if (n_out) n_out[b] = nc;
        if (n > W - 1 && overflow) *overflow = 1;
    }
    for (int u = nc + 1 + lane; u < W; u += 64) {
        yi[u] = 0;
        yo[u] = 0;
    }
}

// Decoder_CIF.preprocess (src/transformer/decoder.py:356-366) + the decoder's lengths in one launch: ys_in[b, u] = (<sos>, target[b, :-1])[u]
// where target[b, u] > 0, else 0; in_len[b] = number of positive targets.  One wave per utterance.
__global__ __launch_bounds__(256) void decoder_cif_targets_kernel(const int64_t* __restrict__ target, int64_t* __restrict__ ys_in,
                                                                  int32_t* __restrict__ in_len, int B, int U, int64_t sos) {
    const int lane = threadIdx.x & 63, b = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (b >= B) return;
    const int64_t* row = target + (int64_t)b * U;
    int n = 0;
    for (int u0 = 0; u0 < U; u0 += 64) {
        const int u = u0 + lane;
        const bool in = u < U;
        const int64_t v = in ? row[u] : 0;
        const int64_t prev = !in ? 0 : (u == 0 ? sos : row[u - 1]);
        if (in) ys_in[(int64_t)b * U + u] = v > 0 ? prev : 0;
        n += __popcll(__ballot(v > 0));
    }
    if (lane == 0 && in_len) in_len[b] = n;
}

}  // namespace

extern "C" int asr_add_layernorm_fwd(void* stream, const float* x, const float* residual, const float* gamma, const float* beta,
                                     const float* pe, const int32_t* row_len, float* y32, void* y16, float* mean, float* rstd,
                                     float* s_out, int B, int L, int D, float eps, asr_dropout_t drop_x, asr_dropout_t drop_y) {
    ASR_REQUIRE(x && gamma && beta && y32, ASR_ERR_ARG, "layernorm: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && D > 0 && D <= 256 * LN_MAXJ && D % 4 == 0, ASR_ERR_UNSUPPORTED,
                "layernorm: D=%d must be a multiple of 4 and <= %d", D, 256 * LN_MAXJ);
    ASR_REQUIRE(drop_x.thr16 < 65536u && drop_y.thr16 < 65536u, ASR_ERR_ARG, "layernorm: dropout thr16 must be < 65536");
    ASR_REQUIRE(asr_aligned(x, 16) && asr_aligned(y32, 16) && asr_aligned(gamma, 16) && asr_aligned(beta, 16) &&
                    (!residual || asr_aligned(residual, 16)) && (!pe || asr_aligned(pe, 16)) && (!y16 || asr_aligned(y16, 8)),
                ASR_ERR_ALIGN, "layernorm: 16-byte alignment required");
    const int M = B * L;
    hipLaunchKernelGGL(add_layernorm_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, residual,
                       gamma, beta, pe, row_len, y32, reinterpret_cast<bf16_t*>(y16), mean, rstd, s_out, M, L, D, eps, drop_x, drop_y);
    ASR_LAUNCH_CHECK("add_layernorm_fwd");
    return 0;
}

extern "C" int asr_decoder_targets(void* stream, const int64_t* targets, int64_t* ys_in, int64_t* ys_out, int32_t* in_len, int64_t* n_out,
                                   int32_t* overflow, int B, int U, int W, int64_t sos_id, int64_t eos_id) {
    ASR_REQUIRE(targets && ys_in && ys_out, ASR_ERR_ARG, "decoder_targets: null pointer");
    ASR_REQUIRE(B > 0 && U > 0 && W >= 1, ASR_ERR_ARG, "decoder_targets: B=%d U=%d W=%d", B, U, W);
    hipLaunchKernelGGL(decoder_targets_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), targets, ys_in, ys_out,
                       in_len, n_out, overflow, B, U, W, sos_id, eos_id);
    ASR_LAUNCH_CHECK("decoder_targets");
    return 0;
}

extern "C" int asr_decoder_cif_targets(void* stream, const int64_t* target, int64_t* ys_in, int32_t* in_len, int B, int U, int64_t sos_id) {
    ASR_REQUIRE(target && ys_in && B > 0 && U > 0, ASR_ERR_ARG, "decoder_cif_targets: bad args");
    hipLaunchKernelGGL(decoder_cif_targets_kernel, dim3((B + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), target, ys_in, in_len,
                       B, U, sos_id);
    ASR_LAUNCH_CHECK("decoder_cif_targets");
    return 0;
}

extern "C" int asr_embed_pe_fwd(void* stream, const int64_t* ids, const float* emb, const float* pe, float* y32, void* y16, int B,
                                int U, int D, int V, asr_dropout_t drop) {
    ASR_REQUIRE(drop.thr16 < 65536u, ASR_ERR_ARG, "embed: dropout thr16 must be < 65536");
    ASR_REQUIRE(ids && emb && pe && y32, ASR_ERR_ARG, "embed: null pointer");
    ASR_REQUIRE(B > 0 && U > 0 && D > 0 && D % 4 == 0 && V > 0, ASR_ERR_ARG, "embed: bad sizes");
    ASR_REQUIRE(asr_aligned(emb, 16) && asr_aligned(pe, 16) && asr_aligned(y32, 16), ASR_ERR_ALIGN, "embed: alignment");
    const int M = B * U;
    hipLaunchKernelGGL(embed_pe_fwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), ids, emb, pe, y32,
                       reinterpret_cast<bf16_t*>(y16), M, U, D, V, drop);
    ASR_LAUNCH_CHECK("embed_pe_fwd");
    return 0;
}

extern "C" int asr_dropout_apply(void* stream, const float* x, float* y, int N0, int N1, int N2, asr_dropout_t drop) {
    ASR_REQUIRE(x && y && N0 > 0 && N1 > 0 && N2 > 0, ASR_ERR_ARG, "dropout_apply: bad args");
    ASR_REQUIRE(drop.thr16 > 0 && drop.thr16 < 65536u, ASR_ERR_ARG, "dropout_apply: thr16 must be in (0, 65536)");
    const int64_t ppn = (int64_t)N1 * ((N2 + 1) / 2);
    ASR_REQUIRE(ppn < (int64_t)1 << 32, ASR_ERR_UNSUPPORTED, "dropout_apply: N1*N2/2 must be < 2^32");
    const int64_t total = ppn * N0;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 4096) blocks = 4096;
    hipLaunchKernelGGL(dropout_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x, y, N1, N2,
                       ppn, total, drop);
    ASR_LAUNCH_CHECK("dropout_apply");
    return 0;
}

extern "C" int asr_cast_f32_bf16(void* stream, const float* x, void* y, int64_t n) {
    ASR_REQUIRE(x && y && n > 0, ASR_ERR_ARG, "cast: bad args");
    ASR_REQUIRE(asr_aligned(x, 16) && asr_aligned(y, 8), ASR_ERR_ALIGN, "cast: alignment");
    const int64_t n4 = n / 4;
    int64_t blocks = (n4 + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    if (blocks < 1) blocks = 1;
    hipLaunchKernelGGL(cast_f32_bf16_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), x,
                       reinterpret_cast<bf16_t*>(y), n4, n);
    ASR_LAUNCH_CHECK("cast_f32_bf16");
    return 0;
}

extern "C" int asr_mask_rows(void* stream, float* x, const int32_t* len, int B, int L, int V, int64_t ld) {
    ASR_REQUIRE(x && len && B > 0 && L > 0 && V > 0 && ld >= V, ASR_ERR_ARG, "mask_rows: bad args");
    hipLaunchKernelGGL(mask_rows_kernel, dim3(B * L), dim3(256), 0, static_cast<hipStream_t>(stream), x, len, L, V, ld);
    ASR_LAUNCH_CHECK("mask_rows");
    return 0;
}

extern "C" int asr_assigner_tail_fwd(void* stream, const float* x, const float* w, const float* b, const int32_t* len, int B, int L,
                                     int Dh, float* alpha) {
    ASR_REQUIRE(x && w && b && len && alpha && B > 0 && L > 0 && Dh > 0, ASR_ERR_ARG, "assigner_tail: bad args");
    const int M = B * L;
    hipLaunchKernelGGL(assigner_tail_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, w, b, len, M, L,
                       Dh, alpha);
    ASR_LAUNCH_CHECK("assigner_tail");
    return 0;
}
