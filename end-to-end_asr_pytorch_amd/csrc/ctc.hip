// Fused CTC loss for gfx950: logits -> log-softmax -> alpha/beta -> loss and d(loss)/d(logits).
// Replaces F.log_softmax + F.ctc_loss at loss.py:41-43 / ctcModel/loss.py:9-11 (aten `_ctc_loss`, Graves 2006).
//
// HBM-bound design: the [B,L,V] logits are streamed exactly once forward (row log-sum-exp + gather
// of the 2U+1 extended-label log-probs into a compact [B,L,S] table) and once backward (softmax recomputed from the
// saved row lse; gradient written once).  The T-long alpha / beta recursions never touch the V axis: they run on the
// compact table, one workgroup per utterance, the previous row held in LDS (log-space, -inf aware), the table rows
// prefetched into registers 4 steps ahead so the dependent chain is LDS + transcendental latency only.
// Per-label occupancies are scattered into an LDS vector indexed by vocabulary id, so repeated labels need no global
// atomics and the gradient row is produced by one coalesced stream.
#include "asr_common.h"

namespace {

__device__ __forceinline__ float lse3(float a, float b, float c) {
    const float m = fmaxf(fmaxf(a, b), c);
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m) + expf(c - m));
}
__device__ __forceinline__ float lse2(float a, float b) {
    const float m = fmaxf(a, b);
    if (m == -INFINITY) return -INFINITY;
    return m + logf(expf(a - m) + expf(b - m));
}

__global__ void ctc_prep_kernel(const int64_t* __restrict__ targets, int B, int Umax, int32_t* __restrict__ tgt_len) {
    const int b = blockIdx.x * blockDim.x + threadIdx.x;
    if (b >= B) return;
    int n = 0;
    for (int u = 0; u < Umax; ++u) n += (targets[(int64_t)b * Umax + u] != 0);  // loss.py:40  targets.ne(0).sum(1)
    tgt_len[b] = n;
}

// one workgroup per (b,t) row: online (max,sum-exp) over V, then gather the extended labels' log-probs
__global__ __launch_bounds__(256) void ctc_lse_gather_kernel(const float* __restrict__ logits, int64_t ldl,
                                                             const int32_t* __restrict__ in_len, const int64_t* __restrict__ targets,
                                                             const int32_t* __restrict__ tgt_len, int L, int V, int Umax, int blank,
                                                             float* __restrict__ lse_out, float* __restrict__ lp_ext) {
    const int row = blockIdx.x;
    const int b = row / L, t = row - b * L;
    if (t >= in_len[b]) return;
    const float* x = logits + (int64_t)row * ldl;
    const int tid = threadIdx.x;
    float m = -INFINITY, s = 0.f;
    const int mis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
    const int peel = min((4 - mis) & 3, V);
    if (tid < peel) { m = x[tid]; s = 1.f; }
    const int nv4 = (V - peel) >> 2;
    const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
    for (int i = tid; i < nv4; i += 256) {
        const f32x4 v = x4[i];
        const float m4 = fmaxf(fmaxf(v[0], v[1]), fmaxf(v[2], v[3]));
        if (m4 > m) { s *= __expf(m - m4); m = m4; }
        s += (__expf(v[0] - m) + __expf(v[1] - m)) + (__expf(v[2] - m) + __expf(v[3] - m));
    }
    const int tail0 = peel + nv4 * 4;
    if (tid < V - tail0) lse_combine(m, s, x[tail0 + tid], 1.f);
    // wave then block combine
#pragma unroll
    for (int o = 32; o > 0; o >>= 1) {
        const float m2 = __shfl_xor(m, o, 64), s2 = __shfl_xor(s, o, 64);
        lse_combine(m, s, m2, s2);
    }
    __shared__ float sm[4], ss[4];
    __shared__ float lse_sh;
    if ((tid & 63) == 0) { sm[tid >> 6] = m; ss[tid >> 6] = s; }
    __syncthreads();
    if (tid == 0) {
        float M = sm[0], S = ss[0];
        for (int w = 1; w < 4; ++w) lse_combine(M, S, sm[w], ss[w]);
        const float l = M + logf(S);
        lse_sh = l;
        lse_out[row] = l;
    }
    __syncthreads();
    const float lse = lse_sh;
    const int Sfull = 2 * Umax + 1, Sb = 2 * tgt_len[b] + 1;
    for (int sidx = tid; sidx < Sfull; sidx += 256) {
        float v = -INFINITY;
        if (sidx < Sb) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            v = x[lab] - lse;
        }
        lp_ext[(int64_t)row * Sfull + sidx] = v;
    }
}

// alpha recursion: one workgroup per utterance, thread s owns extended state s.
template <bool BACKWARD>
__global__ void ctc_recursion_kernel(const float* __restrict__ lp_ext, const int32_t* __restrict__ in_len,
                                     const int64_t* __restrict__ targets, const int32_t* __restrict__ tgt_len, int L, int Umax,
                                     float* __restrict__ alpha, float* __restrict__ nll) {
    extern __shared__ float sh[];  // 2 x (NTH + 4)
    const int b = blockIdx.x, s = threadIdx.x, NTH = blockDim.x;
    const int Sfull = 2 * Umax + 1;
    const int U = tgt_len[b], Sb = 2 * U + 1;
    const int Tb = min(in_len[b], L);
    const bool live = s < Sb;
    float* buf0 = sh + 2;             // index -2..NTH+1 valid
    float* buf1 = sh + (NTH + 4) + 2;
    if (s < 2) { buf0[-2 + s] = -INFINITY; buf1[-2 + s] = -INFINITY; buf0[NTH + s] = -INFINITY; buf1[NTH + s] = -INFINITY; }

    // skip transitions: forward uses s-2 -> s (ext[s] != blank, ext[s] != ext[s-2]); backward uses s -> s+2.
    bool skip = false;
    if (live && (s & 1)) {
        const int64_t* tg = targets + (int64_t)b * Umax;
        if (!BACKWARD) skip = (s >= 3) && (tg[s >> 1] != tg[(s >> 1) - 1]);
        else skip = (s + 2 < Sb) && (tg[s >> 1] != tg[(s >> 1) + 1]);
    }
    const float* lp = lp_ext + (int64_t)b * L * Sfull + (live ? s : 0);
    float* al = alpha + (int64_t)b * L * Sfull + (live ? s : 0);

    if (Tb <= 0) {
        if (!BACKWARD && s == 0) nll[b] = (Sb == 1) ? 0.f : INFINITY;
        return;
    }
    const float my_nll = BACKWARD ? nll[b] : 0.f;
    float a;
    if (!BACKWARD) {
        a = (live && s < 2) ? lp[0] : -INFINITY;
        if (live) al[0] = a;
    } else {
        const float l0 = live ? lp[(int64_t)(Tb - 1) * Sfull] : -INFINITY;
        a = (live && s >= Sb - 2) ? l0 : -INFINITY;
        if (live) {
            float* ap = al + (int64_t)(Tb - 1) * Sfull;
            *ap = __expf(*ap + a - l0 + my_nll);  // occupancy = exp(alpha + beta - lp + nll)
        }
    }
    buf0[s] = a;
    __syncthreads();
    float* cur = buf0;
    float* nxt = buf1;

    // register prefetch of the compact table, 4 time steps per group
    float pf[4], pa[4];
    auto fetch = [&](int step0) {
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = step0 + i;  // step index 1..Tb-1
            const bool ok = live && st < Tb;
            const int t = BACKWARD ? (Tb - 1 - st) : st;
            pf[i] = ok ? lp[(int64_t)t * Sfull] : -INFINITY;
            if (BACKWARD) pa[i] = ok ? al[(int64_t)t * Sfull] : 0.f;
        }
    };
    fetch(1);
    for (int st0 = 1; st0 < Tb; st0 += 4) {
        float cf[4], ca[4];
#pragma unroll
        for (int i = 0; i < 4; ++i) { cf[i] = pf[i]; ca[i] = pa[i]; }
        if (st0 + 4 < Tb) fetch(st0 + 4);
#pragma unroll
        for (int i = 0; i < 4; ++i) {
            const int st = st0 + i;
            if (st < Tb) {  // uniform across the workgroup
                const float a0 = cur[s];
                const float a1 = BACKWARD ? cur[s + 1] : cur[s - 1];
                const float a2 = skip ? (BACKWARD ? cur[s + 2] : cur[s - 2]) : -INFINITY;
                float v = live ? lse3(a0, a1, a2) + cf[i] : -INFINITY;
                nxt[s] = v;
                if (live) {
                    const int t = BACKWARD ? (Tb - 1 - st) : st;
                    if (!BACKWARD) al[(int64_t)t * Sfull] = v;
                    else al[(int64_t)t * Sfull] = __expf(ca[i] + v - cf[i] + my_nll);
                }
                __syncthreads();
                float* tmp = cur; cur = nxt; nxt = tmp;
            }
        }
    }
    if (!BACKWARD && s == 0) {
        const float ll = (Sb > 1) ? lse2(cur[Sb - 1], cur[Sb - 2]) : cur[0];
        nll[b] = -ll;
    }
}

__global__ void ctc_mean_kernel(const float* __restrict__ nll, const int32_t* __restrict__ tgt_len, int B, float* __restrict__ loss) {
    float s = 0.f;
    for (int b = threadIdx.x; b < B; b += 64) s += nll[b] / (float)max(tgt_len[b], 1);
    s = wave_sum(s);
    if (threadIdx.x == 0) loss[0] = s / (float)B;
}

// gradient stream: grid (RB, B); each workgroup walks rows t = blockIdx.x, += gridDim.x of utterance b.
__global__ __launch_bounds__(256) void ctc_grad_kernel(const float* __restrict__ logits, int64_t ldl, const int32_t* __restrict__ in_len,
                                                       const int64_t* __restrict__ targets, const int32_t* __restrict__ tgt_len,
                                                       int B, int L, int V, int Umax, int blank, const float* __restrict__ lse,
                                                       const float* __restrict__ occ, const float* __restrict__ gout,
                                                       float* __restrict__ grad, int64_t ldg) {
    extern __shared__ float corr[];  // V floats
    const int b = blockIdx.y, tid = threadIdx.x;
    const int Sfull = 2 * Umax + 1, Sb = 2 * tgt_len[b] + 1;
    const int Tb = min(in_len[b], L);
    const float scale = gout[0] / ((float)B * (float)max(tgt_len[b], 1));
    for (int i = tid; i < V; i += 256) corr[i] = 0.f;
    __syncthreads();
    for (int t = blockIdx.x; t < L; t += gridDim.x) {
        const int64_t row = (int64_t)b * L + t;
        float* g = grad + row * ldg;
        const int gmis = (int)((reinterpret_cast<uintptr_t>(g) >> 2) & 3);
        if (t >= Tb) {  // padded frame: zero gradient, no reads
            const int peel = min((4 - gmis) & 3, V);
            if (tid < peel) g[tid] = 0.f;
            const int nv4 = (V - peel) >> 2;
            f32x4* g4 = reinterpret_cast<f32x4*>(g + peel);
            for (int i = tid; i < nv4; i += 256) g4[i] = f32x4{0, 0, 0, 0};
            const int tail0 = peel + nv4 * 4;
            if (tid < V - tail0) g[tail0 + tid] = 0.f;
            continue;
        }
        const float* x = logits + row * ldl;
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            atomicAdd(&corr[lab], occ[row * Sfull + sidx]);
        }
        __syncthreads();
        const float l = lse[row];
        const int xmis = (int)((reinterpret_cast<uintptr_t>(x) >> 2) & 3);
        if (xmis == gmis) {
            const int peel = min((4 - gmis) & 3, V);
            if (tid < peel) g[tid] = scale * (__expf(x[tid] - l) - corr[tid]);
            const int nv4 = (V - peel) >> 2;
            const f32x4* x4 = reinterpret_cast<const f32x4*>(x + peel);
            f32x4* g4 = reinterpret_cast<f32x4*>(g + peel);
            for (int i = tid; i < nv4; i += 256) {
                const f32x4 v = x4[i];
                const int c = peel + i * 4;
                f32x4 o;
#pragma unroll
                for (int j = 0; j < 4; ++j) o[j] = scale * (__expf(v[j] - l) - corr[c + j]);
                g4[i] = o;
            }
            const int tail0 = peel + nv4 * 4;
            if (tid < V - tail0) g[tail0 + tid] = scale * (__expf(x[tail0 + tid] - l) - corr[tail0 + tid]);
        } else {
            for (int c = tid; c < V; c += 256) g[c] = scale * (__expf(x[c] - l) - corr[c]);
        }
        __syncthreads();
        for (int sidx = tid; sidx < Sb; sidx += 256) {
            const int lab = (sidx & 1) ? (int)targets[(int64_t)b * Umax + (sidx >> 1)] : blank;
            corr[lab] = 0.f;
        }
        __syncthreads();
    }
}

int recursion_threads(int Umax) {
    const int S = 2 * Umax + 1;
    return ((S + 63) / 64) * 64;
}

}  // namespace

extern "C" int asr_ctc_loss_fwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, float* lse, float* lp_ext, float* alpha, float* nll,
                                int32_t* tgt_len) {
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len, ASR_ERR_ARG, "ctc_fwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && blank >= 0 && blank < V && ldl >= V, ASR_ERR_ARG, "ctc_fwd: bad sizes");
    const int nth = recursion_threads(Umax);
    ASR_REQUIRE(nth <= 1024, ASR_ERR_UNSUPPORTED, "ctc_fwd: Umax=%d too long (2U+1 must be <= 1024)", Umax);
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ctc_prep_kernel, dim3((B + 63) / 64), dim3(64), 0, s, targets, B, Umax, tgt_len);
    hipLaunchKernelGGL(ctc_lse_gather_kernel, dim3(B * L), dim3(256), 0, s, logits, ldl, in_len, targets, tgt_len, L, V, Umax, blank,
                       lse, lp_ext);
    hipLaunchKernelGGL(ctc_recursion_kernel<false>, dim3(B), dim3(nth), 2 * (nth + 4) * sizeof(float), s, lp_ext, in_len, targets,
                       tgt_len, L, Umax, alpha, nll);
    ASR_LAUNCH_CHECK("ctc_loss_fwd");
    return 0;
}

extern "C" int asr_ctc_mean(void* stream, const float* nll, const int32_t* tgt_len, int B, float* loss) {
    ASR_REQUIRE(nll && tgt_len && loss && B > 0, ASR_ERR_ARG, "ctc_mean: bad args");
    hipLaunchKernelGGL(ctc_mean_kernel, dim3(1), dim3(64), 0, static_cast<hipStream_t>(stream), nll, tgt_len, B, loss);
    ASR_LAUNCH_CHECK("ctc_mean");
    return 0;
}

extern "C" int asr_ctc_loss_bwd(void* stream, const float* logits, int64_t ldl, const int32_t* in_len, const int64_t* targets, int B,
                                int L, int V, int Umax, int blank, const float* lse, const float* lp_ext, float* alpha,
                                const float* nll, const int32_t* tgt_len, const float* gout, float* grad, int64_t ldg) {
    ASR_REQUIRE(logits && in_len && targets && lse && lp_ext && alpha && nll && tgt_len && gout && grad, ASR_ERR_ARG,
                "ctc_bwd: null pointer");
    ASR_REQUIRE(B > 0 && L > 0 && V > 1 && Umax > 0 && ldl >= V && ldg >= V, ASR_ERR_ARG, "ctc_bwd: bad sizes");
    ASR_REQUIRE((size_t)V * sizeof(float) <= 64 * 1024, ASR_ERR_UNSUPPORTED, "ctc_bwd: V=%d exceeds the LDS occupancy vector", V);
    const int nth = recursion_threads(Umax);
    ASR_REQUIRE(nth <= 1024, ASR_ERR_UNSUPPORTED, "ctc_bwd: Umax too long");
    hipStream_t s = static_cast<hipStream_t>(stream);
    hipLaunchKernelGGL(ctc_recursion_kernel<true>, dim3(B), dim3(nth), 2 * (nth + 4) * sizeof(float), s, lp_ext, in_len, targets,
                       tgt_len, L, Umax, alpha, const_cast<float*>(nll));
    int rb = (2048 + B - 1) / B;  // ~2048 workgroups in flight
    if (rb > L) rb = L;
    if (rb < 1) rb = 1;
    hipLaunchKernelGGL(ctc_grad_kernel, dim3(rb, B), dim3(256), (size_t)V * sizeof(float), s, logits, ldl, in_len, targets, tgt_len, B, L,
                       V, Umax, blank, lse, alpha, gout, grad, ldg);
    ASR_LAUNCH_CHECK("ctc_loss_bwd");
    return 0;
}
