// Conv2dSubsample (conv_encoder.py:101-108) for gfx950.
//
// Layout: activations between conv layers are channel-LAST [B,T,F,32] so the 32 input channels of one tap are a
// contiguous 64-byte (bf16) / 128-byte (f32) MFMA K-slice.  Only the region of each layer that the final crop
// (freq[:ceil(D/2)], time[:ceil(T/2^n)]) can see is produced: the reference computes 86 frequency columns in the
// last layer and keeps 40.
//   layer 0 (1 -> 32 ch, K = 9): VALU, one thread per output position, weights in LDS, implicit zero right-pad.
//   layer i >= 1 (32 -> 32 ch): implicit GEMM on MFMA, K = 9 taps x 32 channels.  Weights (A operand, rows =
//     c_out) are gathered once per wave into registers; the B operand (16 consecutive output positions x 32 c_in)
//     is read straight from global/L2 per tap - neighbouring taps/positions hit the same lines in L1.
//     D[c_out][position] puts 4 consecutive output channels of one position in each lane.
#include "asr_common.h"

namespace {

template <typename CT>
__global__ __launch_bounds__(256) void conv_sub0_kernel(const float* __restrict__ feats, const float* __restrict__ w0,
                                                        const float* __restrict__ b0, CT* __restrict__ y, int B, int T, int D,
                                                        int T1, int F1) {
    __shared__ float ws[9][32];
    __shared__ float bs[32];
    for (int i = threadIdx.x; i < 288; i += 256) ws[i % 9][i / 9] = w0[i];  // w0[c][0][kh][kw] -> ws[tap][c]
    if (threadIdx.x < 32) bs[threadIdx.x] = b0[threadIdx.x];
    __syncthreads();
    const int64_t total = (int64_t)B * T1 * F1;
    const int64_t pos = (int64_t)blockIdx.x * 256 + threadIdx.x;
    if (pos >= total) return;
    const int f = (int)(pos % F1);
    const int t1 = (int)((pos / F1) % T1);
    const int b = (int)(pos / ((int64_t)F1 * T1));
    float x[9];
#pragma unroll
    for (int kh = 0; kh < 3; ++kh)
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int tt = 2 * t1 + kh, ff = f + kw;
            x[kh * 3 + kw] = (tt < T && ff < D) ? feats[((int64_t)b * T + tt) * D + ff] : 0.f;
        }
    CT* yp = y + pos * 32;
#pragma unroll
    for (int c0 = 0; c0 < 32; c0 += 4) {
        f32x4 o;
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            float acc = bs[c0 + i];
#pragma unroll
            for (int tap = 0; tap < 9; ++tap) acc = fmaf(ws[tap][c0 + i], x[tap], acc);
            o[i] = fmaxf(acc, 0.f);
        }
        if constexpr (sizeof(CT) == 4) {
            *reinterpret_cast<f32x4*>(yp + c0) = o;
        } else {
            bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
            *reinterpret_cast<bf16x4*>(yp + c0) = ob;
        }
    }
}

template <typename CT> struct ConvK {  // 16-byte chunks per 32-channel tap slice
    static constexpr int CH = 16 / sizeof(CT);      // elements per chunk
    static constexpr int GROUPS = 32 / (4 * CH);    // 4-chunk MFMA groups per tap: bf16 1, f32 2
};

template <typename CT>
__global__ __launch_bounds__(256) void conv_sub1_kernel(const CT* __restrict__ x, const float* __restrict__ w,
                                                        const float* __restrict__ bias, CT* __restrict__ y, int B, int Tin,
                                                        int Fin, int Tout, int Fout, int last, int n_tiles) {
    constexpr int CH = ConvK<CT>::CH, G = ConvK<CT>::GROUPS;
    const int lane = threadIdx.x & 63;
    const int r16 = lane & 15, q4 = lane >> 4;
    const int gwave = blockIdx.x * 4 + (threadIdx.x >> 6);
    const int nwaves = gridDim.x * 4;

    // weights as the MFMA row operand: wf[tap][ct][g] = chunk (g*4+q4) of w[c_out = ct*16 + r16][c_in][tap]
    u32x4 wf[9][2][G];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ct = 0; ct < 2; ++ct)
#pragma unroll
            for (int g = 0; g < G; ++g) {
                const int co = ct * 16 + r16;
                const int ci0 = (g * 4 + q4) * CH;
                if constexpr (sizeof(CT) == 4) {
                    f32x4 v;
#pragma unroll
                    for (int j = 0; j < 4; ++j) v[j] = w[(co * 32 + ci0 + j) * 9 + tap];
                    wf[tap][ct][g] = __builtin_bit_cast(u32x4, v);
                } else {
                    bf16x8 v;
#pragma unroll
                    for (int j = 0; j < 8; ++j) v[j] = (bf16_t)w[(co * 32 + ci0 + j) * 9 + tap];
                    wf[tap][ct][g] = __builtin_bit_cast(u32x4, v);
                }
            }
    f32x4 bv[2];
#pragma unroll
    for (int ct = 0; ct < 2; ++ct) bv[ct] = *reinterpret_cast<const f32x4*>(bias + ct * 16 + q4 * 4);

    const int64_t total = (int64_t)B * Tout * Fout;
    for (int tile = gwave; tile < n_tiles; tile += nwaves) {
        int64_t pos = (int64_t)tile * 16 + r16;
        const bool ok = pos < total;
        if (!ok) pos = total - 1;
        const int f = (int)(pos % Fout);
        const int t = (int)((pos / Fout) % Tout);
        const int b = (int)(pos / ((int64_t)Fout * Tout));
        f32x4 acc[2] = {f32x4{0, 0, 0, 0}, f32x4{0, 0, 0, 0}};
#pragma unroll
        for (int kh = 0; kh < 3; ++kh)
#pragma unroll
            for (int kw = 0; kw < 3; ++kw) {
                const CT* xp = x + (((int64_t)b * Tin + 2 * t + kh) * Fin + f + kw) * 32;
#pragma unroll
                for (int g = 0; g < G; ++g) {
                    const u32x4 xv = *reinterpret_cast<const u32x4*>(xp + (g * 4 + q4) * CH);
                    Mma<CT>::run(wf[kh * 3 + kw][0][g], xv, acc[0]);
                    Mma<CT>::run(wf[kh * 3 + kw][1][g], xv, acc[1]);
                }
            }
        if (!ok) continue;
#pragma unroll
        for (int ct = 0; ct < 2; ++ct) {
            f32x4 o = acc[ct] + bv[ct];
#pragma unroll
            for (int i = 0; i < 4; ++i) o[i] = fmaxf(o[i], 0.f);
            const int c0 = ct * 16 + q4 * 4;
            if (last) {
                CT* yp = y + ((int64_t)b * Tout + t) * (32 * Fout) + f;
#pragma unroll
                for (int i = 0; i < 4; ++i) yp[(int64_t)(c0 + i) * Fout] = from_f32<CT>(o[i]);
            } else {
                CT* yp = y + pos * 32 + c0;
                if constexpr (sizeof(CT) == 4) {
                    *reinterpret_cast<f32x4*>(yp) = o;
                } else {
                    bf16x4 ob = {(bf16_t)o[0], (bf16_t)o[1], (bf16_t)o[2], (bf16_t)o[3]};
                    *reinterpret_cast<bf16x4*>(yp) = ob;
                }
            }
        }
    }
}

// ---- backward helpers: the conv layers' gradients are expressed as GEMMs over an explicit patch matrix ------------------
// im2col: col[(b,t,f), tap*C + c] = x[b, 2t+kh, f+kw, c] (zero outside [Tin) x [Fin)); columns 9*C..ldc-1 are zeroed.
template <typename TI, typename TO>
__global__ __launch_bounds__(256) void conv_im2col_kernel(const TI* __restrict__ x, TO* __restrict__ col, int C, int Tin, int Fin,
                                                          int Tout, int Fout, int ldc, int64_t total) {
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < total; i += stride) {
        const int j = (int)(i % ldc);
        const int64_t pos = i / ldc;
        float v = 0.f;
        if (j < 9 * C) {
            const int tap = j / C, c = j - tap * C, kh = tap / 3, kw = tap - kh * 3;
            const int f = (int)(pos % Fout), t = (int)((pos / Fout) % Tout), b = (int)(pos / ((int64_t)Fout * Tout));
            const int tt = 2 * t + kh, ff = f + kw;
            if (tt < Tin && ff < Fin) v = to_f32(x[(((int64_t)b * Tin + tt) * Fin + ff) * C + c]);
        }
        col[i] = from_f32<TO>(v);
    }
}

// bf16 -> bf16 with C % 8 == 0 and ldc % 8 == 0: one thread per 16-byte chunk (8 channels of one tap) - the element-wise kernel above
// spent its time in 64-bit div / mod per 2-byte store (0.6 TB/s on the 184 MB patch matrix of the second conv layer)
__global__ __launch_bounds__(256) void conv_im2col_vec8_kernel(const bf16_t* __restrict__ x, bf16_t* __restrict__ col, int C, int Tin, int Fin,
                                                               int Tout, int Fout, int ldc8, int64_t npos) {
    const int cpt = C >> 3;                                        // chunks per tap
    const int j8 = threadIdx.x % ldc8, prow = threadIdx.x / ldc8, rows_per_block = 256 / ldc8;
    const int tap = j8 / cpt, c8 = (j8 - tap * cpt) * 8, kh = tap / 3, kw = tap - kh * 3;
    const bool live = prow < rows_per_block && tap < 9;
    for (int64_t pos = (int64_t)blockIdx.x * rows_per_block + prow; pos < npos && prow < rows_per_block; pos += (int64_t)gridDim.x * rows_per_block) {
        u32x4 v = {0u, 0u, 0u, 0u};
        if (live) {
            const int f = (int)(pos % Fout);
            const int64_t bt = pos / Fout;
            const int t = (int)(bt % Tout), b = (int)(bt / Tout);
            const int tt = 2 * t + kh, ff = f + kw;
            if (tt < Tin && ff < Fin) v = *reinterpret_cast<const u32x4*>(x + (((int64_t)b * Tin + tt) * Fin + ff) * C + c8);
        }
        *reinterpret_cast<u32x4*>(col + pos * (int64_t)ldc8 * 8 + j8 * 8) = v;
    }
}

// col2im (gather form) + ReLU mask: dx[b,ti,fi,c] = (y[b,ti,fi,c] > 0) * sum_{kh,kw} dcol[(b,(ti-kh)/2,fi-kw), tap*32 + c]
// over taps with (ti-kh) even, 0 <= (ti-kh)/2 < Tout, 0 <= fi-kw < Fout.  One thread per (position, 4 channels).
template <typename T>
__global__ __launch_bounds__(256) void conv_col2im_kernel(const T* __restrict__ dcol, int ldc, const T* __restrict__ y,
                                                          T* __restrict__ dx, int Tin, int Fin, int Tout, int Fout, int64_t total) {
    typedef T vec4 __attribute__((ext_vector_type(4)));
    const int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= total) return;
    const int c4 = (int)(i & 7) * 4;
    const int64_t pos = i >> 3;
    const int fi = (int)(pos % Fin), ti = (int)((pos / Fin) % Tin), b = (int)(pos / ((int64_t)Fin * Tin));
    f32x4 acc = {0, 0, 0, 0};
#pragma unroll
    for (int kh = 0; kh < 3; ++kh) {
        const int t2 = ti - kh;
        if (t2 < 0 || (t2 & 1)) continue;
        const int t = t2 >> 1;
        if (t >= Tout) continue;
#pragma unroll
        for (int kw = 0; kw < 3; ++kw) {
            const int f = fi - kw;
            if (f < 0 || f >= Fout) continue;
            const vec4 v = *reinterpret_cast<const vec4*>(dcol + (((int64_t)b * Tout + t) * Fout + f) * ldc + (kh * 3 + kw) * 32 + c4);
            acc += f32x4{(float)v[0], (float)v[1], (float)v[2], (float)v[3]};
        }
    }
    const vec4 yv = *reinterpret_cast<const vec4*>(y + pos * 32 + c4);
    vec4 o;
#pragma unroll
    for (int e = 0; e < 4; ++e) o[e] = (T)(((float)yv[e] > 0.f) ? acc[e] : 0.f);
    *reinterpret_cast<vec4*>(dx + pos * 32 + c4) = o;
}

}  // namespace

extern "C" int asr_conv_im2col(void* stream, const void* x, int x_dtype, int C, void* col, int col_dtype, int ldc, int B, int Tin, int Fin,
                               int Tout, int Fout) {
    ASR_REQUIRE(x && col && B > 0 && C > 0 && ldc >= 9 * C && Tout > 0 && Fout > 0, ASR_ERR_ARG, "im2col: bad args");
    const int64_t total = (int64_t)B * Tout * Fout * ldc;
    int64_t blocks = (total + 255) / 256;
    if (blocks > 8192) blocks = 8192;
    hipStream_t s = static_cast<hipStream_t>(stream);
    dim3 g((unsigned)blocks), b(256);
    if (x_dtype == ASR_F32 && col_dtype == ASR_F32)
        hipLaunchKernelGGL((conv_im2col_kernel<float, float>), g, b, 0, s, (const float*)x, (float*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else if (x_dtype == ASR_F32 && col_dtype == ASR_BF16)
        hipLaunchKernelGGL((conv_im2col_kernel<float, bf16_t>), g, b, 0, s, (const float*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else if (x_dtype == ASR_BF16 && col_dtype == ASR_BF16 && C % 8 == 0 && ldc % 8 == 0 && ldc / 8 <= 256 && asr_aligned(x, 16) && asr_aligned(col, 16)) {
        const int ldc8 = ldc / 8, rpb = 256 / ldc8;
        const int64_t npos = (int64_t)B * Tout * Fout;
        int64_t nb = (npos + rpb - 1) / rpb;
        if (nb > 16384) nb = 16384;
        hipLaunchKernelGGL(conv_im2col_vec8_kernel, dim3((unsigned)nb), b, 0, s, (const bf16_t*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc8, npos);
    } else if (x_dtype == ASR_BF16 && col_dtype == ASR_BF16)
        hipLaunchKernelGGL((conv_im2col_kernel<bf16_t, bf16_t>), g, b, 0, s, (const bf16_t*)x, (bf16_t*)col, C, Tin, Fin, Tout, Fout, ldc, total);
    else
        ASR_REQUIRE(false, ASR_ERR_UNSUPPORTED, "im2col: dtype combination");
    ASR_LAUNCH_CHECK("conv_im2col");
    return 0;
}

extern "C" int asr_conv_col2im_relu(void* stream, const void* dcol, int ldc, const void* y, void* dx, int B, int Tin, int Fin, int Tout,
                                    int Fout) {
    ASR_REQUIRE(dcol && y && dx && B > 0 && ldc >= 288 && ldc % 4 == 0, ASR_ERR_ARG, "col2im: bad args");
    const int64_t total = (int64_t)B * Tin * Fin * 8;
    hipLaunchKernelGGL(conv_col2im_kernel<bf16_t>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream),
                       (const bf16_t*)dcol, ldc, (const bf16_t*)y, (bf16_t*)dx, Tin, Fin, Tout, Fout, total);
    ASR_LAUNCH_CHECK("conv_col2im");
    return 0;
}

extern "C" int asr_conv_col2im_relu_f32(void* stream, const float* dcol, int ldc, const float* y, float* dx, int B, int Tin, int Fin, int Tout,
                                        int Fout) {
    ASR_REQUIRE(dcol && y && dx && B > 0 && ldc >= 288 && ldc % 4 == 0, ASR_ERR_ARG, "col2im_f32: bad args");
    const int64_t total = (int64_t)B * Tin * Fin * 8;
    hipLaunchKernelGGL(conv_col2im_kernel<float>, dim3((unsigned)((total + 255) / 256)), dim3(256), 0, static_cast<hipStream_t>(stream), dcol,
                       ldc, y, dx, Tin, Fin, Tout, Fout, total);
    ASR_LAUNCH_CHECK("conv_col2im_f32");
    return 0;
}

extern "C" int asr_conv_sub0_fwd(void* stream, const float* feats, const float* w0, const float* b0, void* y, int dtype, int B, int T,
                                 int D, int T1, int F1) {
    ASR_REQUIRE(feats && w0 && b0 && y, ASR_ERR_ARG, "conv_sub0: null pointer");
    ASR_REQUIRE(B > 0 && T > 0 && D > 0 && T1 > 0 && F1 > 0, ASR_ERR_ARG, "conv_sub0: bad sizes");
    ASR_REQUIRE(asr_aligned(y, 16), ASR_ERR_ALIGN, "conv_sub0: y alignment");
    const int64_t total = (int64_t)B * T1 * F1;
    dim3 grid((unsigned)((total + 255) / 256)), block(256);
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ASR_F32)
        hipLaunchKernelGGL(conv_sub0_kernel<float>, grid, block, 0, s, feats, w0, b0, (float*)y, B, T, D, T1, F1);
    else
        hipLaunchKernelGGL(conv_sub0_kernel<bf16_t>, grid, block, 0, s, feats, w0, b0, (bf16_t*)y, B, T, D, T1, F1);
    ASR_LAUNCH_CHECK("conv_sub0");
    return 0;
}

extern "C" int asr_conv_sub1_fwd(void* stream, const void* x, const float* w, const float* b, void* y, int dtype, int B, int Tin,
                                 int Fin, int Tout, int Fout, int last) {
    ASR_REQUIRE(x && w && b && y, ASR_ERR_ARG, "conv_sub1: null pointer");
    ASR_REQUIRE(B > 0 && Tout > 0 && Fout > 0 && Tin >= 2 * Tout + 1 && Fin >= Fout + 2, ASR_ERR_ARG,
                "conv_sub1: input region [%d,%d] too small for output [%d,%d]", Tin, Fin, Tout, Fout);
    ASR_REQUIRE(asr_aligned(x, 16) && asr_aligned(y, 16) && asr_aligned(b, 16), ASR_ERR_ALIGN, "conv_sub1: alignment");
    const int64_t total = (int64_t)B * Tout * Fout;
    const int n_tiles = (int)((total + 15) / 16);
    int blocks = (n_tiles + 3) / 4;
    if (blocks > 1024) blocks = 1024;
    hipStream_t s = static_cast<hipStream_t>(stream);
    if (dtype == ASR_F32)
        hipLaunchKernelGGL(conv_sub1_kernel<float>, dim3(blocks), dim3(256), 0, s, (const float*)x, w, b, (float*)y, B, Tin, Fin,
                           Tout, Fout, last, n_tiles);
    else
        hipLaunchKernelGGL(conv_sub1_kernel<bf16_t>, dim3(blocks), dim3(256), 0, s, (const bf16_t*)x, w, b, (bf16_t*)y, B, Tin, Fin,
                           Tout, Fout, last, n_tiles);
    ASR_LAUNCH_CHECK("conv_sub1");
    return 0;
}
