// Greedy decoding helpers: what sits between the logits and the token ids in
//   Decoder.step / batch_decode   src/transformer/decoder.py:98-164   F.log_softmax, torch.argmax over the vocabulary
//   GreedyDecoder.decode          src/ctcModel/ctc_infer.py:28-46,69-80   torch.max over the vocabulary, collapse repeats, drop blanks
// All HBM-bound row kernels: one wavefront per row / per utterance.
#include "asr_common.h"

namespace {

// index of the row maximum; ties go to the lowest index (torch.argmax / torch.max on CPU)
__global__ __launch_bounds__(256) void argmax_rows_kernel(const float* __restrict__ x, int64_t ld, int M, int V, int64_t* __restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ld;
    float best = -INFINITY;
    int bi = 0x7fffffff;
    for (int c = lane; c < V; c += 64) {
        const float v = xr[c];
        if (v > best || bi == 0x7fffffff) { best = v; bi = c; }      // strictly greater: the first maximum of this lane's columns
    }
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float v2 = __shfl_xor(best, o, 64);
        const int i2 = __shfl_xor(bi, o, 64);
        if (v2 > best || (v2 == best && i2 < bi)) { best = v2; bi = i2; }
    }
    if (lane == 0) out[row] = bi == 0x7fffffff ? 0 : bi;
}

// y = x - logsumexp(x) per row (F.log_softmax, decoder.py:118)
__global__ __launch_bounds__(256) void log_softmax_rows_kernel(const float* __restrict__ x, int64_t ldx, int M, int V, float* __restrict__ y,
                                                               int64_t ldy) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const float* xr = x + row * ldx;
    float m = -INFINITY, s = 0.f;
    for (int c = lane; c < V; c += 64) lse_combine(m, s, xr[c], 1.f);
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    const float l = m + logf(s);
    for (int c = lane; c < V; c += 64) y[row * ldy + c] = xr[c] - l;
}

// CTC collapse (ctc_infer.py:37-46): walk the first len[b] frame labels, keep a label that is not blank and differs from the
// previous FRAME's label.  One wavefront per utterance, 64 frames per step, compaction by ballot + popcount.
__global__ __launch_bounds__(64) void ctc_greedy_reduce_kernel(const int64_t* __restrict__ frames, const int32_t* __restrict__ len, int L,
                                                               int blank, int64_t* __restrict__ out, int32_t* __restrict__ out_len) {
    const int b = blockIdx.x, lane = threadIdx.x;
    const int64_t* f = frames + (int64_t)b * L;
    int64_t* o = out + (int64_t)b * L;
    const int n = min(len[b], L);
    int kept = 0;
    for (int t0 = 0; t0 < n; t0 += 64) {
        const int t = t0 + lane;
        const int64_t tok = t < n ? f[t] : blank;
        const int64_t prev = (t > 0 && t < n) ? f[t - 1] : -1;
        const bool keep = t < n && tok != blank && tok != prev;
        const unsigned long long mask = __ballot(keep);
        if (keep) o[kept + __builtin_popcountll(mask & ((1ull << lane) - 1ull))] = tok;
        kept += __builtin_popcountll(mask);
    }
    for (int t = kept + lane; t < L; t += 64) o[t] = 0;      // zero padding (padding_list_seqs pad=0)
    if (lane == 0) out_len[b] = kept;
}

}  // namespace

extern "C" int asr_argmax_rows(void* stream, const float* x, int64_t ld, int M, int V, int64_t* out) {
    ASR_REQUIRE(x && out && M > 0 && V > 0 && ld >= V, ASR_ERR_ARG, "argmax_rows: bad args");
    hipLaunchKernelGGL(argmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, ld, M, V, out);
    ASR_LAUNCH_CHECK("argmax_rows");
    return 0;
}

extern "C" int asr_log_softmax_rows(void* stream, const float* x, int64_t ldx, int M, int V, float* y, int64_t ldy) {
    ASR_REQUIRE(x && y && M > 0 && V > 0 && ldx >= V && ldy >= V, ASR_ERR_ARG, "log_softmax_rows: bad args");
    hipLaunchKernelGGL(log_softmax_rows_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), x, ldx, M, V, y, ldy);
    ASR_LAUNCH_CHECK("log_softmax_rows");
    return 0;
}

extern "C" int asr_ctc_greedy_reduce(void* stream, const int64_t* frames, const int32_t* len, int B, int L, int blank, int64_t* out,
                                     int32_t* out_len) {
    ASR_REQUIRE(frames && len && out && out_len && B > 0 && L > 0, ASR_ERR_ARG, "ctc_greedy_reduce: bad args");
    hipLaunchKernelGGL(ctc_greedy_reduce_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), frames, len, L, blank, out, out_len);
    ASR_LAUNCH_CHECK("ctc_greedy_reduce");
    return 0;
}
