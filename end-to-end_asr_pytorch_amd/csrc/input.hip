// The input step in front of the path, on the device:
//   lfr_stack    low-frame-rate stacking: src/utils/data.py:191-218 (build_LFR_features: stack m frames, skip n, the last
//                frame repeated past the end of the utterance), for a whole padded batch at once
//   spec_aug     src/utils/utils.py:168-194: frequency bands replaced by each frame's mean over frequency, time spans by the
//                utterance's mean over time, both means taken from the UNmasked features; the reference loops over masks and over
//                utterances in Python (B x masks slice assignments), here it is two passes: statistics, then one elementwise pass
//                that decides per element which mask - if any - wrote it last
// HBM-bound copies / reductions; one wavefront per row.
#include "asr_common.h"

namespace {

__global__ __launch_bounds__(256) void lfr_stack_kernel(const float* __restrict__ x, const int32_t* __restrict__ len, int T, int D, int m,
                                                        int n, int Tl, int64_t rows, float* __restrict__ y, int32_t* __restrict__ len_out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);      // output row (b, i)
    if (row >= rows) return;
    const int b = (int)(row / Tl), i = (int)(row - (int64_t)b * Tl);
    const int Tb = min(len[b], T);
    const int Tlb = (Tb + n - 1) / n;                                      // ceil(T / n) (data.py:207)
    if (i == 0 && lane == 0) len_out[b] = Tlb;
    float* yr = y + row * (int64_t)m * D;
    if (i >= Tlb) {
        for (int c = lane; c < m * D; c += 64) yr[c] = 0.f;
        return;
    }
    for (int j = 0; j < m; ++j) {
        const int t = min(i * n + j, Tb - 1);                              // the last frame stands in for the missing ones (:212-216)
        const float* xr = x + ((int64_t)b * T + t) * D;
        for (int c = lane; c < D; c += 64) yr[j * D + c] = xr[c];
    }
}

// statistics of the unmasked features: fmean[b,t] = mean over frequency; tsum[b,v] += x[b,t,v] (caller-zeroed; divided by the length
// in the apply pass: utils.py:171-173, "features are padded with zeros")
__global__ __launch_bounds__(256) void spec_aug_stats_kernel(const float* __restrict__ x, int M, int T, int V, float* __restrict__ fmean,
                                                             float* __restrict__ tsum) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = (int)(row / T);
    float s = 0.f;
    for (int c = lane; c < V; c += 64) {
        const float v = x[row * V + c];
        s += v;
        if (v != 0.f) atomicAdd(tsum + (int64_t)b * V + c, v);
    }
    s = wave_sum(s);
    if (lane == 0) fmean[row] = s / (float)V;
}

// r: uniform [0,1) draws in the reference's order - for every mask loop iteration k (first the frequency masks, then the time
// masks) rand(B) for the width and rand(B) for the start: r[(2k + which) * B + b].
//   width = (long)(max_width * r_w)            start = (long)((float)(extent - width) * r_s)      (utils.py:179-182,187-190)
// x[b, s:s+w] as Python slices it: a negative bound counts from the END of the padded axis (an utterance shorter than the drawn
// time-mask width gives start = (long)((len - w) * r) < 0 - utils.py:189-192 then writes a run ending at the padded T, or
// nothing when the end wraps differently), bounds past the axis are clamped
__device__ __forceinline__ bool in_py_slice(int64_t i, int64_t s, int64_t w, int64_t n) {
    int64_t lo = s, hi = s + w;
    if (lo < 0) lo = lo + n < 0 ? 0 : lo + n;
    if (hi < 0) hi = hi + n < 0 ? 0 : hi + n;
    if (hi > n) hi = n;
    return i >= lo && i < hi;
}

__global__ __launch_bounds__(256) void spec_aug_apply_kernel(float* __restrict__ x, const int32_t* __restrict__ len, int B, int T, int V,
                                                             const float* __restrict__ fmean, const float* __restrict__ tsum,
                                                             const float* __restrict__ r, int n_freq, int freq_w, int n_time, int time_w) {
    const int64_t total = (int64_t)B * T * V;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int v = (int)(i % V);
        const int64_t bt = i / V;
        const int t = (int)(bt % T), b = (int)(bt / T);
        bool in_time = false, in_freq = false;
        const float lb = (float)len[b];
        for (int k = 0; k < n_time; ++k) {
            const int64_t ts = (int64_t)((float)time_w * r[(int64_t)(2 * (n_freq + k)) * B + b]);
            const int64_t t0 = (int64_t)(((float)(len[b] - ts)) * r[(int64_t)(2 * (n_freq + k) + 1) * B + b]);
            in_time = in_time || in_py_slice(t, t0, ts, T);
        }
        for (int k = 0; k < n_freq; ++k) {
            const int64_t fs = (int64_t)((float)freq_w * r[(int64_t)(2 * k) * B + b]);
            const int64_t f0 = (int64_t)(((float)(V - fs)) * r[(int64_t)(2 * k + 1) * B + b]);
            in_freq = in_freq || in_py_slice(v, f0, fs, V);
        }
        // time masks are written after frequency masks (utils.py:176-192): a time mask wins where both cover an element
        if (in_time) x[i] = tsum[(int64_t)b * V + v] / lb;
        else if (in_freq) x[i] = fmean[bt];
    }
}

// mask_lm token masking (src/mask_lm/Mask_LM.py:19-41): keep[b,t] = AND_{j=0..M} (r[b,(t+j) mod T] > p) - the reference ANDs the
// draw mask with M circular left shifts of itself, so a draw <= p blanks the M positions before it as well; masked tokens become 0.
__global__ __launch_bounds__(256) void token_mask_kernel(const int64_t* __restrict__ ids, const float* __restrict__ r, int B, int T, float p,
                                                         int M, int64_t* __restrict__ out, unsigned char* __restrict__ masked) {
    const int64_t total = (int64_t)B * T;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (int64_t)gridDim.x * 256) {
        const int b = (int)(i / T), t = (int)(i - (int64_t)b * T);
        bool keep = true;
        for (int j = 0; j <= M; ++j) keep = keep && (r[(int64_t)b * T + (t + j) % T] > p);
        out[i] = keep ? ids[i] : 0;
        masked[i] = keep ? 0 : 1;
    }
}

}  // namespace

extern "C" int asr_token_mask(void* stream, const int64_t* ids, const float* rand01, int B, int T, float p, int M, int64_t* out,
                              unsigned char* masked) {
    ASR_REQUIRE(ids && rand01 && out && masked && B > 0 && T > 0 && M >= 0, ASR_ERR_ARG, "token_mask: bad args");
    int64_t blocks = ((int64_t)B * T + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(token_mask_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream), ids, rand01, B, T, p, M, out,
                       masked);
    ASR_LAUNCH_CHECK("token_mask");
    return 0;
}

extern "C" int asr_lfr_stack(void* stream, const float* x, const int32_t* len, int B, int T, int D, int m, int n, float* y,
                             int32_t* len_out) {
    ASR_REQUIRE(x && len && y && len_out && B > 0 && T > 0 && D > 0 && m > 0 && n > 0, ASR_ERR_ARG, "lfr_stack: bad args");
    const int Tl = (T + n - 1) / n;
    const int64_t rows = (int64_t)B * Tl;
    hipLaunchKernelGGL(lfr_stack_kernel, dim3((unsigned)((rows + 3) / 4)), dim3(256), 0, static_cast<hipStream_t>(stream), x, len, T, D, m, n,
                       Tl, rows, y, len_out);
    ASR_LAUNCH_CHECK("lfr_stack");
    return 0;
}

extern "C" int asr_spec_aug(void* stream, float* x, const int32_t* len, int B, int T, int V, const float* rand01, int n_freq, int freq_width,
                            int n_time, int time_width, float* fmean, float* tsum) {
    ASR_REQUIRE(x && len && rand01 && fmean && tsum && B > 0 && T > 0 && V > 0 && n_freq >= 0 && n_time >= 0, ASR_ERR_ARG,
                "spec_aug: bad args");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipError_t e = hipMemsetAsync(tsum, 0, (size_t)B * V * sizeof(float), s);
    if (e != hipSuccess) {
        asr_set_error("spec_aug: %s", hipGetErrorString(e));
        return (int)e;
    }
    const int M = B * T;
    hipLaunchKernelGGL(spec_aug_stats_kernel, dim3((M + 3) / 4), dim3(256), 0, s, x, M, T, V, fmean, tsum);
    int64_t blocks = ((int64_t)M * V + 255) / 256;
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(spec_aug_apply_kernel, dim3((unsigned)blocks), dim3(256), 0, s, x, len, B, T, V, fmean, tsum, rand01, n_freq, freq_width,
                       n_time, time_width);
    ASR_LAUNCH_CHECK("spec_aug");
    return 0;
}
