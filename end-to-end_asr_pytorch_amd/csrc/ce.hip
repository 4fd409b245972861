// Label-smoothed cross entropy (loss.py:5-31) for gfx950: one workgroup per logits row, single streaming pass
// (online max / sum-exp / plain sum), un-normalised smoothing exactly as the reference writes it:
//   w_c = (1-eps) for the target, eps/V elsewhere (weights do NOT sum to 1);  loss_row = -sum_c w_c * log_softmax_c.
// Pad rows (target == 0) contribute nothing and the mean divides by the number of non-pad rows.
#include "asr_common.h"

namespace {

__global__ __launch_bounds__(256) void ce_fwd_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                     int V, float eps, float* __restrict__ row_loss, float* __restrict__ lse_out) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (int64_t)row * ldl;
    float m = -INFINITY, s = 0.f, sum = 0.f;
    for (int c = tid; c < V; c += 256) {
        const float v = x[c];
        sum += v;
        if (v > m) { s *= __expf(m - v); m = v; }
        s += __expf(v - m);
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
        sum += __shfl_xor(sum, o, 64);
    }
    __shared__ float sm[4], ss[4], sx[4];
    if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; sx[tid >> 6] = sum; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0], X = sx[0];
        for (int w = 1; w < 4; ++w) { lse_combine(M, S, sm[w], ss[w]); X += sx[w]; }
        const float lse = M + logf(S);
        lse_out[row] = lse;
        const int64_t tg = targets[row];
        float loss = 0.f;
        if (tg != 0) {
            const float lp_t = x[tg] - lse;
            if (eps > 0.f) {
                const float sum_lp = X - (float)V * lse;
                loss = -((1.f - eps) * lp_t + (eps / (float)V) * (sum_lp - lp_t));
            } else {
                loss = -lp_t;
            }
        }
        row_loss[row] = loss;
    }
}

// `counted` (optional, one byte per row): the denominator counts only rows with counted[i] != 0 while the numerator still sums
// every non-pad row - mask_lm's cal_ce_mask_loss (src/mask_lm/loss.py:24-30)
__global__ __launch_bounds__(256) void ce_mean_kernel(const float* __restrict__ row_loss, const int64_t* __restrict__ targets,
                                                      const unsigned char* __restrict__ counted, int N, float* __restrict__ out) {
    float s = 0.f, n = 0.f;
    for (int i = threadIdx.x; i < N; i += 256) {
        const bool np = targets[i] != 0;
        s += np ? row_loss[i] : 0.f;
        n += (np && (!counted || counted[i])) ? 1.f : 0.f;
    }
    s = wave_sum(s);
    n = wave_sum(n);
    __shared__ float a[4], c[4];
    if ((threadIdx.x & 63) == 0) { a[threadIdx.x >> 6] = s; c[threadIdx.x >> 6] = n; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const float S = (a[0] + a[1]) + (a[2] + a[3]), Nw = (c[0] + c[1]) + (c[2] + c[3]);
        out[0] = S / Nw;
        out[1] = Nw;
    }
}

// GT = float: columns 0 .. V-1 are written.  GT = bf16_t (the trainer's gradient image, see asr_ctc_loss_bwd): the whole row 0 .. ldg-1
// is written, zeros past V, so ldg can be the backward GEMMs' padded row width.
template <typename GT>
__global__ __launch_bounds__(256) void ce_bwd_kernel(const float* __restrict__ logits, int64_t ldl, const int64_t* __restrict__ targets,
                                                     int V, float eps, const float* __restrict__ lse, const float* __restrict__ n_word,
                                                     const float* __restrict__ gout, GT* __restrict__ grad, int64_t ldg) {
    const int row = blockIdx.x, tid = threadIdx.x;
    const float* x = logits + (int64_t)row * ldl;
    GT* g = grad + (int64_t)row * ldg;
    const int64_t tg = targets[row];
    const int W = sizeof(GT) == 2 ? (int)ldg : V;
    for (int c = V + tid; c < W; c += 256) g[c] = (GT)0.f;
    if (tg == 0) {
        for (int c = tid; c < V; c += 256) g[c] = (GT)0.f;
        return;
    }
    const float gs = gout[0] / n_word[0];
    const float l = lse[row];
    const float w_off = (eps > 0.f) ? eps / (float)V : 0.f;
    const float w_on = (eps > 0.f) ? (1.f - eps) : 1.f;
    const float wsum = w_on + (float)(V - 1) * w_off;
    for (int c = tid; c < V; c += 256) {
        const float p = __expf(x[c] - l);
        g[c] = (GT)(gs * (p * wsum - ((c == tg) ? w_on : w_off)));
    }
}

}  // namespace

extern "C" int asr_ce_loss_fwd(void* stream, const float* logits, int64_t ldl, const int64_t* targets, int N, int V, float smoothing,
                               float* row_loss, float* lse) {
    ASR_REQUIRE(logits && targets && row_loss && lse && N > 0 && V > 0 && ldl >= V, ASR_ERR_ARG, "ce_fwd: bad args");
    hipLaunchKernelGGL(ce_fwd_kernel, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ldl, targets, V, smoothing,
                       row_loss, lse);
    ASR_LAUNCH_CHECK("ce_loss_fwd");
    return 0;
}

extern "C" int asr_ce_mean(void* stream, const float* row_loss, const int64_t* targets, int N, float* loss) {
    ASR_REQUIRE(row_loss && targets && loss && N > 0, ASR_ERR_ARG, "ce_mean: bad args");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), row_loss, targets, nullptr, N, loss);
    ASR_LAUNCH_CHECK("ce_mean");
    return 0;
}

extern "C" int asr_ce_mean_masked(void* stream, const float* row_loss, const int64_t* targets, const unsigned char* counted, int N,
                                  float* loss) {
    ASR_REQUIRE(row_loss && targets && counted && loss && N > 0, ASR_ERR_ARG, "ce_mean_masked: bad args");
    hipLaunchKernelGGL(ce_mean_kernel, dim3(1), dim3(256), 0, static_cast<hipStream_t>(stream), row_loss, targets, counted, N, loss);
    ASR_LAUNCH_CHECK("ce_mean_masked");
    return 0;
}

extern "C" int asr_ce_loss_bwd(void* stream, const float* logits, int64_t ldl, const int64_t* targets, int N, int V, float smoothing,
                               const float* lse, const float* n_word, const float* gout, void* grad, int grad_dtype, int64_t ldg) {
    ASR_REQUIRE(logits && targets && lse && n_word && gout && grad && N > 0 && V > 0 && ldg >= V, ASR_ERR_ARG, "ce_bwd: bad args");
    ASR_REQUIRE(grad_dtype == ASR_F32 || grad_dtype == ASR_BF16, ASR_ERR_ARG, "ce_bwd: bad grad_dtype");
    if (grad_dtype == ASR_BF16)
        hipLaunchKernelGGL(ce_bwd_kernel<bf16_t>, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ldl, targets, V, smoothing,
                           lse, n_word, gout, reinterpret_cast<bf16_t*>(grad), ldg);
    else
        hipLaunchKernelGGL(ce_bwd_kernel<float>, dim3(N), dim3(256), 0, static_cast<hipStream_t>(stream), logits, ldl, targets, V, smoothing, lse,
                           n_word, gout, reinterpret_cast<float*>(grad), ldg);
    ASR_LAUNCH_CHECK("ce_loss_bwd");
    return 0;
}
