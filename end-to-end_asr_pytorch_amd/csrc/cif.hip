// CIF - continuous integrate-and-fire (cif_model.py:57-106) for gfx950.
//
// The firing decision `integrate > threshold` is a discrete function of an fp32 running sum that is decremented by
// exactly 1.0 on fire; reproducing the reference's boundaries requires the reference's operation ORDER, so the
// recurrence itself is not re-associated.  One wavefront per utterance walks the T frames 64 at a time: the recurrence runs on
// wave-uniform values (the utterance's weights sit in LDS and are read as broadcasts, 5 vector instructions per frame, ~20 cycles) and only leaves the
// running sum before each frame in that frame's lane; fire flags, cur / rem weights, token indices and the fire list are then
// computed lane-parallel per chunk.  The
// H-wide weighted sums - the actual bandwidth - run in a second, fully parallel kernel: one workgroup per output
// token, threads over channels, frames of the token's segment accumulated in the reference's order with separate
// (non-fused) multiply and add so the fp32 results match the CPU loop bit-for-bit.
#include "asr_common.h"

// The reference rounds the product and the sum separately (torch: cur[:,None]*hidden, then +=); hipcc contracts
// a*b+c into v_fma_f32 by default, which changes the last bit.  Contraction is disabled for this whole file (pragma
// here + -ffp-contract=off in the build; the __fmul_rn/__fadd_rn header inlines keep their own contract flag, so the
// arithmetic below is written with plain operators).
#pragma clang fp contract(off)

namespace {

// One frame of the recurrence on wave-uniform values (cif_model.py:71-77): 4 vector instructions, 3 of them on the dependent
// chain, none through a condition register: fire = clamp((s - thr) * 2^100, 0, 1) is exactly 1.0 when s > thr and 0.0 otherwise
// (the sign of a float difference is exact, and two distinct floats of this magnitude differ by far more than 2^-100), so
// s - fire is the reference's `integrate - 1` on fire and `integrate` otherwise - bit for bit - without the compare -> VCC ->
// select round trip (measured 58 cycles per frame with it).  `al` is the frame's weight broadcast from LDS, `hist` collects in lane i the running sum
// BEFORE frame i (one v_cndmask under a one-hot lane mask that a scalar shift moves along) - everything else a frame needs (its
// own fire decision, cur, rem, token index) is recomputed from it lane-parallel after the chunk, with the same fp32 operations on
// the same values, i.e. bit-identically.
#define CIF_STEP(al)                                                                                                            \
    {                                                                                                                           \
        const float s_ = integrate + (al);                                                                                      \
        asm volatile("v_cndmask_b32 %[h], %[h], %[v], %[m]\n\ts_lshl_b64 %[m], %[m], 1" : [h] "+v"(hist), [m] "+s"(onehot) : [v] "v"(integrate) : "scc"); \
        float f_;                                                                                                               \
        asm("v_fma_f32 %0, %1, %2, %3 clamp" : "=v"(f_) : "v"(s_), "v"(big), "v"(nthr_big));                                    \
        integrate = s_ - f_;                                                                                                    \
    }

// `alphas.sum(-1)` in the order torch's CPU kernel adds (SumKernel.cpp: cascade_sum -> vectorized_inner_sum -> row_sum -> multi_row_sum;
// restated with its derivation in oracle.aten_row_sum_f32 and pinned bit for bit on torch's own sums, fixture G16): cif_model.py:95
// ROUNDS this sum to the label count, so the last bit decides rows whose sum sits within an ulp of k + 0.5.  Four interleaved chains
// of 8-lane vector adds = 32 independent fp32 chains, one per lane (lanes 0..31; rows shorter than 8: 4 scalar chains), a four-level
// cascade inside each, then the chains, the tail elements and the 8 lanes added sequentially.  The row is read straight from
// global memory (it is in L1 / L2: the scan has just read it); no multiply-add in sight, so nothing can be contracted.
__device__ __forceinline__ float aten_row_sum_f32(const float* __restrict__ a, int n, int lane) {
    const int W = n >= 8 ? 8 : 1;
    const int vec = n / W, size_ilp = vec >> 2;
    int lg = 0;
    while ((1 << lg) < size_ilp) ++lg;                      // ceil(log2(size_ilp)), 0 for size_ilp <= 1
    const int lp = max(4, lg / 4), step = 1 << lp, mask = step - 1;
    const bool chain = lane < 4 * W;
    const int k = chain ? lane / W : 0, v = chain ? lane % W : 0;
    float acc0 = 0.f, acc1 = 0.f, acc2 = 0.f, acc3 = 0.f;
    int i = 0;
    while (i + step <= size_ilp) {
        for (int j = 0; j < step; ++j, ++i) acc0 += a[(i * 4 + k) * W + v];
        acc1 += acc0; acc0 = 0.f;
        if ((i & (mask << lp)) == 0) {
            acc2 += acc1; acc1 = 0.f;
            if ((i & (mask << (2 * lp))) == 0) { acc3 += acc2; acc2 = 0.f; }
        }
    }
    for (; i < size_ilp; ++i) acc0 += a[(i * 4 + k) * W + v];
    acc0 += acc1;
    acc0 += acc2;
    acc0 += acc3;
    for (int i2 = size_ilp * 4; i2 < vec; ++i2) {           // leftover vectors go to chain 0
        const float x = a[i2 * W + v];
        if (k == 0) acc0 += x;
    }
    for (int kk = 1; kk < 4; ++kk) {                        // chains 1, 2, 3 onto chain 0, in that order
        const float o = __shfl(acc0, kk * W + v, 64);
        if (k == 0) acc0 += o;
    }
    if (W == 1) return __shfl(acc0, 0, 64);
    float fin = 0.f;
    for (int e = vec * 8; e < n; ++e) fin += a[e];
    for (int vv = 0; vv < 8; ++vv) fin += __shfl(acc0, vv, 64);
    return fin;
}

constexpr int CIF_LDS_FRAMES = 8192;      // frames of one utterance staged in LDS per pass (32 KiB)

__global__ __launch_bounds__(64) void cif_scan_kernel(const float* __restrict__ alpha, int L, float thr, float* __restrict__ cur_out,
                                                      float* __restrict__ rem_out, int32_t* __restrict__ fire_idx,
                                                      int32_t* __restrict__ n_fire, int32_t* __restrict__ n_label,
                                                      int32_t* __restrict__ tok_out) {
    __shared__ __attribute__((aligned(16))) float row[CIF_LDS_FRAMES];
    const int b = blockIdx.x, lane = threadIdx.x;
    const float* __restrict__ a = alpha + (int64_t)b * L;
    float integrate = 0.f;
    int n = 0;
    const float big = 0x1p100f, nthr_big = -thr * 0x1p100f;      // (thr * 2^100 is exact: a power-of-two scaling)
    for (int base = 0; base < L; base += CIF_LDS_FRAMES) {
        const int len = min(CIF_LDS_FRAMES, L - base);
        // the utterance's weights into LDS: 16 independent loads per lane in flight (one round trip per 1024 frames)
        for (int t0 = 0; t0 < len; t0 += 1024) {
            float v[16];
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = t0 + j * 64 + lane;
                const float x = a[base + min(t, len - 1)];      // (clamped: branch-free, all 16 loads in flight)
                v[j] = (t < len) ? x : 0.f;
            }
#pragma unroll
            for (int j = 0; j < 16; ++j) {
                const int t = t0 + j * 64 + lane;
                if (t < len) row[t] = v[j];
            }
        }
        __syncthreads();
        for (int t0 = 0; t0 < len; t0 += 64) {
            const int t = t0 + lane;
            const float av = (t < len) ? row[t] : 0.f;
            float hist = 0.f;
            const int cnt = min(64, len - t0);
            if (cnt == 64) {
                unsigned long long onehot = 1ull;
                const f32x4* r4 = reinterpret_cast<const f32x4*>(row + t0);       // wave-uniform addresses: broadcast reads
                f32x4 q[16];       // all 64 weights into registers first: the chain below then never waits on LDS
#pragma unroll
                for (int i = 0; i < 16; ++i) q[i] = r4[i];
#pragma unroll
                for (int i = 0; i < 16; ++i) { CIF_STEP(q[i][0]) CIF_STEP(q[i][1]) CIF_STEP(q[i][2]) CIF_STEP(q[i][3]) }
            } else {
                for (int i = 0; i < cnt; ++i) {      // ragged tail (< 64 frames): same steps
                    const float al = row[t0 + i];
                    const float s_ = integrate + al;
                    hist = (lane == i) ? integrate : hist;
                    integrate = (s_ > thr) ? (s_ - 1.0f) : s_;
                }
            }
            // lane-parallel: this lane's frame from the sum that stood before it
            const float sm = hist + av;
            const bool fire = (t < len) && (sm > thr);
            const unsigned long long fires = __ballot(fire);
            const int before = __builtin_popcountll(fires & ((1ull << lane) - 1ull));
            const float c = fire ? (1.0f - hist) : av;
            if (t < len) {
                const int64_t o = (int64_t)b * L + base + t;
                cur_out[o] = c;
                rem_out[o] = av - c;
                if (tok_out) tok_out[o] = (n + before) | (fire ? (1 << 30) : 0);
                if (fire) fire_idx[(int64_t)b * L + n + before] = base + t;
            }
            n += __builtin_popcountll(fires);
        }
        __syncthreads();
    }
    // round(sum alpha) (cif_model.py:95 `torch.round(alphas.sum(-1)).int()`): the reference's fp32 sum, in the reference's order
    const float asum = aten_row_sum_f32(a, L, lane);
    if (lane == 0) {
        n_fire[b] = n;
        n_label[b] = (int32_t)rintf(asum);  // torch.round: half to even
    }
}

__global__ __launch_bounds__(256) void cif_gather_kernel(const float* __restrict__ hidden, const float* __restrict__ cur,
                                                         const float* __restrict__ rem, const int32_t* __restrict__ fire_idx,
                                                         const int32_t* __restrict__ n_fire, int L, int H, int Umax,
                                                         float* __restrict__ out) {
    const int u = blockIdx.x % Umax, b = blockIdx.x / Umax;
    float* o = out + ((int64_t)b * Umax + u) * H;
    if (u >= n_fire[b]) {
        for (int c = threadIdx.x; c < H; c += 256) o[c] = 0.f;
        return;
    }
    const int t_end = fire_idx[(int64_t)b * L + u];
    const int t_start = (u == 0) ? 0 : fire_idx[(int64_t)b * L + u - 1];
    const float* hb = hidden + (int64_t)b * L * H;
    const float* cb = cur + (int64_t)b * L;
    const float w0 = (u == 0) ? cb[0] : rem[(int64_t)b * L + t_start];
    for (int c = threadIdx.x; c < H; c += 256) {
        float frame = w0 * hb[(int64_t)t_start * H + c];
        for (int t = t_start + 1; t <= t_end; ++t) {
            const float prod = cb[t] * hb[(int64_t)t * H + c];  // rounded product, then rounded sum (no FMA)
            frame = frame + prod;
        }
        o[c] = frame;
    }
}

// ---- backward ------------------------------------------------------------------------------------------------------------
// out[b,u,:] = sum_t w[u,t] h[b,t,:] with w = cur_t for the frames of token u and rem_f for the fire frame f that opened it.
// One wave per frame (b,t): d_hidden[t] = cur_t * g(tok_t) (+ rem_t * g(tok_t + 1) on a fire frame), d_cur[t] = <g(tok_t), h_t>,
// d_rem[t] = <g(tok_t + 1), h_t>, where g(u) = d_out[b,u,:] for emitted tokens (u < n_fire) and 0 otherwise.
__global__ __launch_bounds__(256) void cif_gather_bwd_kernel(const float* __restrict__ hidden, const float* __restrict__ cur,
                                                             const float* __restrict__ rem, const int32_t* __restrict__ tok,
                                                             const int32_t* __restrict__ n_fire, const float* __restrict__ d_out, int M,
                                                             int L, int H, int Umax, float* __restrict__ d_hidden,
                                                             float* __restrict__ d_cur, float* __restrict__ d_rem) {
    const int lane = threadIdx.x & 63;
    const int64_t row = (int64_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (row >= M) return;
    const int b = (int)(row / L);
    const int te = tok[row];
    const int u = te & 0x3fffffff;
    const bool fire = (te >> 30) & 1;
    const int nf = min(n_fire[b], Umax);
    const bool has1 = u < nf, has2 = fire && (u + 1 < nf);
    const float c = cur[row], rm = rem[row];
    const float* g1 = d_out + ((int64_t)b * Umax + u) * H;
    const float* g2 = g1 + H;
    float s1 = 0.f, s2 = 0.f;
    for (int k = lane * 4; k < H; k += 256) {
        const f32x4 hv = *reinterpret_cast<const f32x4*>(hidden + row * H + k);
        f32x4 o = {0, 0, 0, 0};
        if (has1) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(g1 + k);
            o = a * c;
            s1 += (a[0] * hv[0] + a[1] * hv[1]) + (a[2] * hv[2] + a[3] * hv[3]);
        }
        if (has2) {
            const f32x4 a = *reinterpret_cast<const f32x4*>(g2 + k);
            o += a * rm;
            s2 += (a[0] * hv[0] + a[1] * hv[1]) + (a[2] * hv[2] + a[3] * hv[3]);
        }
        *reinterpret_cast<f32x4*>(d_hidden + row * H + k) = o;
    }
    s1 = wave_sum(s1);
    s2 = wave_sum(s2);
    if (lane == 0) { d_cur[row] = s1; d_rem[row] = s2; }
}

// d_alpha from d_cur / d_rem.  Non-fire frame: cur = alpha_t (d_alpha_t += d_cur_t).  Fire frame f: cur_f = 1 - integrate_before_f,
// rem_f = alpha_f - cur_f, and integrate_before_f = sum_{j<f} alpha_j - (#fires before f): every EARLIER alpha_j receives
// (d_rem_f - d_cur_f), alpha_f itself receives d_rem_f.  So d_alpha_t = own_t + sum_{fires f > t} (d_rem_f - d_cur_f): a suffix sum,
// done by one wave per utterance walking 64-frame chunks from the end.
__global__ __launch_bounds__(64) void cif_scan_bwd_kernel(const float* __restrict__ d_cur, const float* __restrict__ d_rem,
                                                          const int32_t* __restrict__ tok, int L, float* __restrict__ d_alpha) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float carry = 0.f;   // sum of g over all frames after the current chunk
    for (int t0 = ((L - 1) / 64) * 64; t0 >= 0; t0 -= 64) {
        const int t = t0 + lane;
        const bool ok = t < L;
        const int64_t idx = (int64_t)b * L + (ok ? t : 0);
        const bool fire = ok && ((tok[idx] >> 30) & 1);
        const float dc = ok ? d_cur[idx] : 0.f, dr = ok ? d_rem[idx] : 0.f;
        const float g = fire ? (dr - dc) : 0.f;
        float incl = g;                               // inclusive suffix sum within the chunk
#pragma unroll
        for (int o = 1; o < 64; o <<= 1) {
            const float v = __shfl_down(incl, o, 64);
            if (lane + o < 64) incl += v;
        }
        const float excl = incl - g + carry;          // sum over frames strictly after t
        if (ok) d_alpha[idx] = (fire ? dr : dc) + excl;
        carry += __shfl(incl, 0, 64);
    }
}

}  // namespace

extern "C" int asr_cif_gather_bwd(void* stream, const float* hidden, const float* cur, const float* rem, const int32_t* tok,
                                  const int32_t* n_fire, const float* d_out, int B, int L, int H, int Umax, float* d_hidden, float* d_cur,
                                  float* d_rem) {
    ASR_REQUIRE(hidden && cur && rem && tok && n_fire && d_out && d_hidden && d_cur && d_rem, ASR_ERR_ARG, "cif_gather_bwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && H > 0 && H % 4 == 0 && Umax > 0, ASR_ERR_ARG, "cif_gather_bwd: bad sizes");
    const int M = B * L;
    hipLaunchKernelGGL(cif_gather_bwd_kernel, dim3((M + 3) / 4), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, cur, rem, tok, n_fire,
                       d_out, M, L, H, Umax, d_hidden, d_cur, d_rem);
    ASR_LAUNCH_CHECK("cif_gather_bwd");
    return 0;
}

extern "C" int asr_cif_scan_bwd(void* stream, const float* d_cur, const float* d_rem, const int32_t* tok, int B, int L, float* d_alpha) {
    ASR_REQUIRE(d_cur && d_rem && tok && d_alpha && B > 0 && L > 0, ASR_ERR_ARG, "cif_scan_bwd: bad args");
    hipLaunchKernelGGL(cif_scan_bwd_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), d_cur, d_rem, tok, L, d_alpha);
    ASR_LAUNCH_CHECK("cif_scan_bwd");
    return 0;
}

extern "C" int asr_cif_scan_fwd(void* stream, const float* alpha, int B, int L, float threshold, float* cur, float* rem,
                                int32_t* fire_idx, int32_t* n_fire, int32_t* n_label, int32_t* tok) {
    ASR_REQUIRE(alpha && cur && rem && fire_idx && n_fire && n_label && B > 0 && L > 0, ASR_ERR_ARG, "cif_scan: bad args");
    hipLaunchKernelGGL(cif_scan_kernel, dim3(B), dim3(64), 0, static_cast<hipStream_t>(stream), alpha, L, threshold, cur, rem, fire_idx,
                       n_fire, n_label, tok);
    ASR_LAUNCH_CHECK("cif_scan_fwd");
    return 0;
}

extern "C" int asr_cif_gather_fwd(void* stream, const float* hidden, const float* cur, const float* rem, const int32_t* fire_idx,
                                  const int32_t* n_fire, int B, int L, int H, int Umax, float* out) {
    ASR_REQUIRE(hidden && cur && rem && fire_idx && n_fire && out && B > 0 && L > 0 && H > 0 && Umax > 0, ASR_ERR_ARG,
                "cif_gather: bad args");
    hipLaunchKernelGGL(cif_gather_kernel, dim3(B * Umax), dim3(256), 0, static_cast<hipStream_t>(stream), hidden, cur, rem, fire_idx,
                       n_fire, L, H, Umax, out);
    ASR_LAUNCH_CHECK("cif_gather_fwd");
    return 0;
}
