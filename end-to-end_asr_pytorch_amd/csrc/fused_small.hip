// Small-M fused block:  y = LayerNorm( dropout(A . W^T + bias) + residual ) [masked rows zeroed]   for D = 256 outputs.
//
// The decoder works on B*(U+1) = 1632 rows: its output projection / second FFN GEMM followed by the residual + LayerNorm kernel
// are two dependent launches of ~10 us each for ~0.2 / 1.7 GFLOP, and a dependent launch costs ~5 us on this stack whatever it
// does.  Here one workgroup owns 16 complete rows: four waves multiply the 16 x K row block with 64 output
// columns each (MFMA 16x16x32 bf16, operands straight from global / L2 into registers - the weight is at most 1 MiB and every
// workgroup reads all of it, fine for ~100 workgroups, wrong for 2000), then the row statistics are reduced across the waves
// through LDS and the LayerNorm epilogue writes exactly what asr_add_layernorm_fwd would have: pre-norm sum, mean, rstd, y32, y16.
// (attention.py:58-60, module.py:50-52 for decoder-sized inputs; the backward uses the unfused kernels on the saved tensors.)
#include "asr_common.h"

namespace {

template <int NK>   // K-steps of 32 kept in flight
__global__ __launch_bounds__(256) void gemm_ln_small_kernel(const bf16_t* __restrict__ A, int64_t lda, const bf16_t* __restrict__ W,
                                                            const float* __restrict__ bias, const float* __restrict__ res,
                                                            const float* __restrict__ gamma, const float* __restrict__ beta,
                                                            const int32_t* __restrict__ row_len, float* __restrict__ s_out,
                                                            float* __restrict__ y32, bf16_t* __restrict__ y16,
                                                            float* __restrict__ mean_out, float* __restrict__ rstd_out, int M, int L,
                                                            int K, float eps, asr_dropout_t drop_x_in) {
    constexpr int D = 256;
    const asr_dropout_t drop_x = drop_resolve(drop_x_in);
    __shared__ float red[2][16][4];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, r16 = lane & 15, q4 = lane >> 4;
    const int row0 = blockIdx.x * 16;
    const int arow = min(row0 + r16, M - 1);                       // rows past M are clamped (computed, never stored)
    const bf16_t* ap = A + (int64_t)arow * lda + q4 * 8;
    const bf16_t* wp = W + (int64_t)(wave * 64 + r16) * K + q4 * 8;  // + 16 j rows for fragment j
    f32x4 acc[4];
#pragma unroll
    for (int j = 0; j < 4; ++j) acc[j] = f32x4{0, 0, 0, 0};
    // a ring of PD K-steps in registers: a step is 4 MFMAs (64 cycles) against ~1 us of load latency, and ~100 workgroups cannot hide
    // that with occupancy - with one step of prefetch the K = 2048 walk took 64 us
    constexpr int PD = NK;
    u32x4 ar[PD], wr[PD][4];
    auto load = [&](int k0, u32x4& a, u32x4 (&w)[4]) {
        a = *reinterpret_cast<const u32x4*>(ap + k0);
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = *reinterpret_cast<const u32x4*>(wp + (int64_t)16 * j * K + k0);
    };
    // K is a multiple of 32 * PD (host) and the body is branch-free - the reload past the end is clamped to the last K-step instead of
    // skipped: with a conditional reload the compiler's s_waitcnt insertion has to assume the shortest path and drains the whole
    // ring (vmcnt(0)) before the first MFMA of every round
#pragma unroll
    for (int sidx = 0; sidx < PD; ++sidx) load(sidx * 32, ar[sidx], wr[sidx]);
    for (int k0 = 0; k0 < K; k0 += 32 * PD) {
#pragma unroll
        for (int sidx = 0; sidx < PD; ++sidx) {
#pragma unroll
            for (int j = 0; j < 4; ++j) Mma<bf16_t>::run(wr[sidx][j], ar[sidx], acc[j]);   // D[n][m]: lane holds row m = r16, cols 16 j + 4 q4 + x
            load(min(k0 + 32 * (sidx + PD), K - 32), ar[sidx], wr[sidx]);
        }
    }
    // ---- epilogue: bias, dropout, residual, LayerNorm over the 256 columns of row r16 (spread over 4 lanes x 4 waves) ----
    const int row = row0 + r16;
    const bool live = row < M;
    const int rr = live ? row : M - 1;
    const int b = rr / L, t = rr - b * L;
    const uint32_t subx = drop_x.thr16 ? drop_subkey(drop_x, (uint32_t)b) : 0u;
    const float scx = drop_scale(drop_x);
    f32x4 v[4];
    float sum = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wave * 64 + 16 * j + 4 * q4;
        v[j] = acc[j];
        if (bias) v[j] += *reinterpret_cast<const f32x4*>(bias + c);
        if (drop_x.thr16) v[j] = drop4(drop_x, subx, (uint32_t)t, D >> 1, c, v[j], scx);
        if (res) v[j] += *reinterpret_cast<const f32x4*>(res + (int64_t)rr * D + c);
        if (live) *reinterpret_cast<f32x4*>(s_out + (int64_t)row * D + c) = v[j];
        sum += (v[j][0] + v[j][1]) + (v[j][2] + v[j][3]);
    }
    sum += __shfl_xor(sum, 16, 64);
    sum += __shfl_xor(sum, 32, 64);
    if (q4 == 0) red[0][r16][wave] = sum;
    __syncthreads();
    const float mean = ((red[0][r16][0] + red[0][r16][1]) + (red[0][r16][2] + red[0][r16][3])) * (1.f / D);
    float q = 0.f;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const f32x4 d = v[j] - mean;
        q += (d[0] * d[0] + d[1] * d[1]) + (d[2] * d[2] + d[3] * d[3]);
    }
    q += __shfl_xor(q, 16, 64);
    q += __shfl_xor(q, 32, 64);
    if (q4 == 0) red[1][r16][wave] = q;
    __syncthreads();
    const float var = ((red[1][r16][0] + red[1][r16][1]) + (red[1][r16][2] + red[1][r16][3])) * (1.f / D);
    const float rstd = 1.0f / sqrtf(var + eps);
    if (!live) return;
    if (wave == 0 && q4 == 0) {
        if (mean_out) mean_out[row] = mean;
        if (rstd_out) rstd_out[row] = rstd;
    }
    const bool keep = row_len ? (t < row_len[b]) : true;
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const int c = wave * 64 + 16 * j + 4 * q4;
        const f32x4 g = *reinterpret_cast<const f32x4*>(gamma + c);
        const f32x4 bt = *reinterpret_cast<const f32x4*>(beta + c);
        f32x4 o = (v[j] - mean) * rstd * g + bt;
        if (!keep) o = f32x4{0, 0, 0, 0};
        *reinterpret_cast<f32x4*>(y32 + (int64_t)row * D + c) = o;
        if (y16) *reinterpret_cast<bf16x4*>(y16 + (int64_t)row * D + c) = bf16x4{(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
    }
}

}  // namespace

extern "C" int asr_gemm_add_layernorm_small(void* stream, const void* A, int64_t lda, const void* W, const float* bias, const float* residual,
                                            const float* gamma, const float* beta, const int32_t* row_len, float* s_out, float* y32,
                                            void* y16, float* mean, float* rstd, int B, int L, int K, float eps, asr_dropout_t drop_x) {
    ASR_REQUIRE(A && W && gamma && beta && s_out && y32 && B > 0 && L > 0, ASR_ERR_ARG, "gemm_add_layernorm_small: null pointer / bad sizes");
    ASR_REQUIRE(K > 0 && K % 32 == 0 && lda % 8 == 0 && lda >= K, ASR_ERR_ALIGN, "gemm_add_layernorm_small: K=%d must be a multiple of 32, lda a multiple of 8", K);
    ASR_REQUIRE(drop_x.thr16 < 65536u, ASR_ERR_ARG, "gemm_add_layernorm_small: dropout thr16 must be < 65536");
    ASR_REQUIRE(asr_aligned(A, 16) && asr_aligned(W, 16) && asr_aligned(s_out, 16) && asr_aligned(y32, 16) && asr_aligned(gamma, 16) &&
                    asr_aligned(beta, 16) && (!bias || asr_aligned(bias, 16)) && (!residual || asr_aligned(residual, 16)) &&
                    (!y16 || asr_aligned(y16, 8)), ASR_ERR_ALIGN, "gemm_add_layernorm_small: 16-byte alignment");
    const int M = B * L;
#define LAUNCH_SMALL(PD)                                                                                                                  \
    hipLaunchKernelGGL(gemm_ln_small_kernel<PD>, dim3((M + 15) / 16), dim3(256), 0, static_cast<hipStream_t>(stream), (const bf16_t*)A, lda, \
                       (const bf16_t*)W, bias, residual, gamma, beta, row_len, s_out, y32, (bf16_t*)y16, mean, rstd, M, L, K, eps, drop_x)
    if (K % 256 == 0) LAUNCH_SMALL(8);
    else if (K % 64 == 0) LAUNCH_SMALL(2);
    else LAUNCH_SMALL(1);
#undef LAUNCH_SMALL
    ASR_LAUNCH_CHECK("gemm_add_layernorm_small");
    return 0;
}
