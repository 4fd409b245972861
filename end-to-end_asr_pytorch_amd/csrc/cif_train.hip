// Small kernels of the CIF family's training path (cif_model.py:44-48, attentionAssigner.py:25-40, conv_encoder.py:33-49) and
// of the backward tape's bookkeeping: everything that used to be an eager tensor expression between two HIP kernels.
//   add2d              dst += src over a [rows, cols] window of two row-strided f32 buffers (gradient accumulation at a join)
//   add_transposed     dst[o][a][b] += src[o][b][a]: weight gradients the GEMMs produce in their own (tap-major) column order
//   relu_mask_mul      d * (y > 0): ReLU backward from the saved forward output
//   conv1d_overlap_add gradient wrt a conv1d input from the gradient wrt its overlapping row windows
//   assigner_tail_bwd  sigmoid + Linear(d_h -> 1) backward (attentionAssigner.py:37-40)
//   cif_rescale_fwd/bwd  alpha = alpha_raw * (num + noise - 0.5) / sum(alpha_raw)  (cif_model.py:44-48) and its backward
#include "asr_common.h"

namespace {

__global__ __launch_bounds__(256) void add2d_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src, int64_t lds,
                                                    int rows, int cols4) {
    const int64_t total = (int64_t)rows * cols4;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols4;
        const int c = (int)(i - r * cols4) * 4;
        f32x4* d = reinterpret_cast<f32x4*>(dst + r * ldd + c);
        *d = *d + *reinterpret_cast<const f32x4*>(src + r * lds + c);
    }
}

__global__ __launch_bounds__(256) void add2d_scalar_kernel(float* __restrict__ dst, int64_t ldd, const float* __restrict__ src,
                                                           int64_t lds, int rows, int cols) {
    const int64_t total = (int64_t)rows * cols;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cols;
        const int c = (int)(i - r * cols);
        dst[r * ldd + c] += src[r * lds + c];
    }
}

// dst [O][A][B] (contiguous) += src[o * lds + b * A + a]
__global__ __launch_bounds__(256) void add_transposed_kernel(float* __restrict__ dst, const float* __restrict__ src, int O, int A, int Bn,
                                                             int64_t lds) {
    const int64_t total = (int64_t)O * A * Bn;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t o = i / ((int64_t)A * Bn);
        const int rem = (int)(i - o * A * Bn);
        const int a = rem / Bn, b = rem - a * Bn;
        dst[i] += src[o * lds + (int64_t)b * A + a];
    }
}

template <typename TY>
__global__ __launch_bounds__(256) void relu_mask_mul_kernel(const float* __restrict__ d, const TY* __restrict__ y, float* __restrict__ out,
                                                            int64_t n) {
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256)
        out[i] = (to_f32(y[i]) > 0.f) ? d[i] : 0.f;
}

// d_in[r][c] = sum_j d_win[r - j][j * cin + c] over the window slots j whose source row r - j exists (rows_in = rows + w)
__global__ __launch_bounds__(256) void conv1d_overlap_add_kernel(const float* __restrict__ d_win, int rows, int w, int cin,
                                                                 float* __restrict__ d_in) {
    const int64_t total = (int64_t)(rows + w) * cin;
    const int64_t ldw = (int64_t)w * cin;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int64_t r = i / cin;
        const int c = (int)(i - r * cin);
        float s = 0.f;
        for (int j = 0; j < w; ++j) {
            const int64_t rs = r - j;
            if (rs >= 0 && rs < rows) s += d_win[rs * ldw + (int64_t)j * cin + c];
        }
        d_in[i] = s;
    }
}

// z = h . w + b, alpha = sigmoid(z) * mask:  dz = g * alpha * (1 - alpha) (0 on masked frames, where alpha is 0);
// d_h[r][:] = dz_r * w;  dw += sum_r dz_r * h[r][:];  db += sum_r dz_r.   One wave per row, persistent over rows; the wave keeps
// its share of dw in registers (Dh <= 1024) and adds it once at the end.
__global__ __launch_bounds__(256) void assigner_tail_bwd_kernel(const float* __restrict__ g, const float* __restrict__ alpha,
                                                                const float* __restrict__ h, const float* __restrict__ w, int M, int Dh,
                                                                float* __restrict__ d_h, float* __restrict__ dw, float* __restrict__ db) {
    const int lane = threadIdx.x & 63;
    const int wave_id = blockIdx.x * 4 + (threadIdx.x >> 6), nwaves = gridDim.x * 4;
    float acc[16];
#pragma unroll
    for (int k = 0; k < 16; ++k) acc[k] = 0.f;
    float accb = 0.f;
    for (int64_t row = wave_id; row < M; row += nwaves) {
        const float a = alpha[row];
        const float dz = g[row] * a * (1.0f - a);
        accb += dz;
#pragma unroll
        for (int k = 0; k < 16; ++k) {
            const int c = k * 64 + lane;
            if (c < Dh) {
                d_h[row * Dh + c] = dz * w[c];
                acc[k] = fmaf(dz, h[row * Dh + c], acc[k]);
            }
        }
    }
#pragma unroll
    for (int k = 0; k < 16; ++k) {
        const int c = k * 64 + lane;
        if (c < Dh) atomicAdd(dw + c, acc[k]);
    }
    if (lane == 0) atomicAdd(db, accb);
}

// one wave per utterance: _num = sum(alpha_raw) (f64 accumulate), num = count(targets > 0), scale = (num + noise - 0.5) / _num,
// alpha = alpha_raw * scale
__global__ __launch_bounds__(64) void cif_rescale_fwd_kernel(const float* __restrict__ alpha_raw, const int64_t* __restrict__ targets,
                                                             const float* __restrict__ noise, int L, int U, float* __restrict__ alpha,
                                                             float* __restrict__ num_pred, float* __restrict__ num,
                                                             float* __restrict__ scale_out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    double s = 0.0;
    for (int t = lane; t < L; t += 64) s += (double)alpha_raw[(int64_t)b * L + t];
    int cnt = 0;
    for (int u = lane; u < U; u += 64) cnt += targets[(int64_t)b * U + u] > 0 ? 1 : 0;
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        s += __shfl_xor(s, o, 64);
        cnt += __shfl_xor(cnt, o, 64);
    }
    const float np = (float)s, nm = (float)cnt;
    const float sc = ((nm + noise[b]) - 0.5f) / np;
    if (lane == 0) {
        num_pred[b] = np;
        num[b] = nm;
        scale_out[b] = sc;
    }
    for (int t = lane; t < L; t += 64) alpha[(int64_t)b * L + t] = alpha_raw[(int64_t)b * L + t] * sc;
}

// d(alpha_raw)[t] = d_alpha[t] * scale + d_num, with d_num = d_num_in (quantity loss) - sum_t(d_alpha * alpha_raw) * scale / _num
__global__ __launch_bounds__(64) void cif_rescale_bwd_kernel(const float* __restrict__ d_alpha, const float* __restrict__ alpha_raw,
                                                             const float* __restrict__ scale, const float* __restrict__ num_pred,
                                                             const float* __restrict__ d_num_in, int L, float* __restrict__ d_raw) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float s = 0.f;
    for (int t = lane; t < L; t += 64) s = fmaf(d_alpha[(int64_t)b * L + t], alpha_raw[(int64_t)b * L + t], s);
    s = wave_sum(s);
    const float sc = scale[b];
    const float dn = (d_num_in ? d_num_in[b] : 0.f) - s * sc / num_pred[b];
    for (int t = lane; t < L; t += 64) d_raw[(int64_t)b * L + t] = fmaf(d_alpha[(int64_t)b * L + t], sc, dn);
}

inline unsigned grid_for(int64_t n) {
    int64_t b = (n + 255) / 256;
    return (unsigned)(b < 1 ? 1 : (b > 2048 ? 2048 : b));
}

}  // namespace

extern "C" int asr_add2d(void* stream, float* dst, int64_t ldd, const float* src, int64_t lds, int rows, int cols) {
    ASR_REQUIRE(dst && src && rows > 0 && cols > 0 && ldd >= cols && lds >= cols, ASR_ERR_ARG, "add2d: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (cols % 4 == 0 && ldd % 4 == 0 && lds % 4 == 0 && asr_aligned(dst, 16) && asr_aligned(src, 16))
        hipLaunchKernelGGL(add2d_kernel, dim3(grid_for((int64_t)rows * cols / 4)), dim3(256), 0, s, dst, ldd, src, lds, rows, cols / 4);
    else
        hipLaunchKernelGGL(add2d_scalar_kernel, dim3(grid_for((int64_t)rows * cols)), dim3(256), 0, s, dst, ldd, src, lds, rows, cols);
    ASR_LAUNCH_CHECK("add2d");
    return 0;
}

extern "C" int asr_add_transposed(void* stream, float* dst, const float* src, int O, int A, int Bn, int64_t lds) {
    ASR_REQUIRE(dst && src && O > 0 && A > 0 && Bn > 0 && lds >= (int64_t)A * Bn, ASR_ERR_ARG, "add_transposed: bad args");
    hipLaunchKernelGGL(add_transposed_kernel, dim3(grid_for((int64_t)O * A * Bn)), dim3(256), 0, static_cast<hipStream_t>(stream), dst, src,
                       O, A, Bn, lds);
    ASR_LAUNCH_CHECK("add_transposed");
    return 0;
}

extern "C" int asr_relu_mask_mul(void* stream, const float* d, const void* y, int y_dtype, float* out, int64_t n) {
    ASR_REQUIRE(d && y && out && n > 0, ASR_ERR_ARG, "relu_mask_mul: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (y_dtype == ASR_BF16)
        hipLaunchKernelGGL(relu_mask_mul_kernel<bf16_t>, dim3(grid_for(n)), dim3(256), 0, s, d, reinterpret_cast<const bf16_t*>(y), out, n);
    else
        hipLaunchKernelGGL(relu_mask_mul_kernel<float>, dim3(grid_for(n)), dim3(256), 0, s, d, reinterpret_cast<const float*>(y), out, n);
    ASR_LAUNCH_CHECK("relu_mask_mul");
    return 0;
}

extern "C" int asr_conv1d_overlap_add(void* stream, const float* d_win, int rows, int w, int cin, float* d_in) {
    ASR_REQUIRE(d_win && d_in && rows > 0 && w > 0 && cin > 0, ASR_ERR_ARG, "conv1d_overlap_add: bad args");
    hipLaunchKernelGGL(conv1d_overlap_add_kernel, dim3(grid_for((int64_t)(rows + w) * cin)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       d_win, rows, w, cin, d_in);
    ASR_LAUNCH_CHECK("conv1d_overlap_add");
    return 0;
}

extern "C" int asr_assigner_tail_bwd(void* stream, const float* g, const float* alpha, const float* h, const float* w, int B, int L, int Dh,
                                     float* d_h, float* dw, float* db) {
    ASR_REQUIRE(g && alpha && h && w && d_h && dw && db && B > 0 && L > 0 && Dh > 0, ASR_ERR_ARG, "assigner_tail_bwd: bad args");
    ASR_REQUIRE(Dh <= 1024, ASR_ERR_UNSUPPORTED, "assigner_tail_bwd: d_hidden must be <= 1024");
    const int M = B * L;
    int blocks = (M + 31) / 32;          // >= 8 rows per wave: few atomics
    if (blocks > 256) blocks = 256;
    hipLaunchKernelGGL(assigner_tail_bwd_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream), g, alpha, h, w, M, Dh, d_h, dw,
                       db);
    ASR_LAUNCH_CHECK("assigner_tail_bwd");
    return 0;
}

extern "C" int asr_cif_rescale_fwd(void* stream, const float* alpha_raw, const int64_t* targets, const float* noise, int B, int L, int U,
                                   float* alpha, float* num_pred, float* num, float* scale) {
    ASR_REQUIRE(alpha_raw && targets && noise && alpha && num_pred && num && scale && B > 0 && L > 0 && U > 0, ASR_ERR_ARG,
                "cif_rescale_fwd: bad args");
    hipLaunchKernelGGL(cif_rescale_fwd_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), alpha_raw, targets, noise, L, U, alpha,
                       num_pred, num, scale);
    ASR_LAUNCH_CHECK("cif_rescale_fwd");
    return 0;
}

extern "C" int asr_cif_rescale_bwd(void* stream, const float* d_alpha, const float* alpha_raw, const float* scale, const float* num_pred,
                                   const float* d_num_in, int B, int L, float* d_raw) {
    ASR_REQUIRE(d_alpha && alpha_raw && scale && num_pred && d_raw && B > 0 && L > 0, ASR_ERR_ARG, "cif_rescale_bwd: bad args");
    hipLaunchKernelGGL(cif_rescale_bwd_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), d_alpha, alpha_raw, scale, num_pred,
                       d_num_in, L, d_raw);
    ASR_LAUNCH_CHECK("cif_rescale_bwd");
    return 0;
}
